"""The C-ABI shared library loads on a CPU-only box and exports exactly what include/shmgan_hip.h
declares; the ctypes signature table mirrors the header.  No compute call is made here."""
import ctypes as C
from pathlib import Path
import re

import pytest

from shmgan_amd import _lib


def test_header_table_and_exports_agree():
    hdr = _lib.header_functions()
    assert len(hdr) == len(set(hdr)), "duplicate declaration in the header"
    assert set(hdr) == set(_lib.SIGNATURES), (set(hdr) ^ set(_lib.SIGNATURES))
    _lib.build()
    L = _lib.lib()
    for name in hdr:
        assert hasattr(L, name), f"{name} not exported by libshmgan_hip.so"


def test_argument_counts_match_header():
    txt = re.sub(r"/\*.*?\*/", "", _lib.HEADER.read_text(), flags=re.S)
    for m in re.finditer(r"\b(shm_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(_lib.SIGNATURES[name][1]), (name, n, len(_lib.SIGNATURES[name][1]))


def test_version_and_error_string_without_gpu():
    L = _lib.lib()
    assert L.shm_version() >= 100
    # shape errors are detected on the host before any launch: safe without a GPU
    rc = L.shm_conv2d_fwd(None, None, 0, 24, 0, None, None, None, 24, 1, 4, 4, 24, 16, 3, 1, 1.0, _lib.F32, None)
    assert rc == -1 and b"null pointer" in L.shm_last_error()
    rc = L.shm_conv2d_wgrad(1, None, 0, 16, 0, 1, 16, 1, 1, 4, 4, 16, 16, 16, 5, 1, 0, 1, 0, _lib.F32, None)
    assert rc == -1 and b"ksize" in L.shm_last_error()
    rc = L.shm_conv2d_wgrad(None, None, 0, 16, 0, None, 16, None, 1, 4, 4, 16, 16, 16, 3, 1, 0, None, 0, _lib.F32, None)
    assert rc == -1 and b"null pointer" in L.shm_last_error()
    # an unknown element type is SHM_E_DTYPE (-2), also detected before any launch
    rc = L.shm_conv2d_wgrad(1, None, 0, 16, 0, 1, 16, 1, 1, 4, 4, 16, 16, 16, 3, 1, 0, 1, 0, 7, None)
    assert rc == -2 and b"dtype" in L.shm_last_error()
    assert L.shm_conv2d_wgrad_workspace(8, 256, 256, 64, 64, 3) > 0
    assert L.shm_image_losses_workspace(8, 256) > 0


def test_product_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from shmgan_amd import ShmGANwithSSpecSeg
    with pytest.raises(RuntimeError):
        ShmGANwithSSpecSeg(image_size=64, filter_size=16)


def test_product_does_not_import_oracle():
    import pathlib
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for p in pathlib.Path(_lib.__file__).resolve().parent.glob("*.py"):
        assert not pat.search(p.read_text()), f"{p} imports the oracle (test infrastructure only)"


def test_trainer_surface_matches_the_reference_object():
    """The methods and signatures main.py / test.py use on the reference's trainer object:
    __init__(args), build_generator(), build_discriminator(), train_step(5 tensors), train(args) (main.py:107),
    custom_per_image_standardization, gram_matrix."""
    import inspect
    from shmgan_amd import ShmGANwithSSpecSeg as T
    assert list(inspect.signature(T.train).parameters)[:2] == ["self", "args"]
    assert list(inspect.signature(T.train_step).parameters)[:6] == ["self", "orig0", "orig45", "orig90", "orig135", "origED"]
    for name in ("build_generator", "build_discriminator", "custom_per_image_standardization", "gram_matrix", "infer",
                 "save_npz", "load_npz"):
        assert callable(getattr(T, name))


def test_tuning_keys_roundtrip_without_gpu():
    L = _lib.lib()
    v = C.c_int(-5)
    assert L.shm_get_tuning(b"tapgemm.variant", C.addressof(v)) == 0 and v.value == 0
    assert L.shm_set_tuning(b"tapgemm.variant", 3) == 0
    assert L.shm_get_tuning(b"tapgemm.variant", C.addressof(v)) == 0 and v.value == 3
    assert L.shm_set_tuning(b"tapgemm.variant", 99) == -1 and b"outside" in L.shm_last_error()
    assert L.shm_set_tuning(b"bogus", 1) == -1 and b"unknown key" in L.shm_last_error()
    assert L.shm_set_tuning(b"reset", 0) == 0
    assert L.shm_get_tuning(b"tapgemm.variant", C.addressof(v)) == 0 and v.value == 0
    for key, dflt in ((b"tapgemm.halo_min_blocks", 1024), (b"tapgemm.small_grid_blocks", 1024), (b"wgrad.variant", 0),
                      (b"wgrad.blocks", 0), (b"stats.fusion", 1), (b"elem.reverse", 1), (b"tapgemm.phase4_min_blocks", 256),
                      (b"wgrad.bf16_rows", 0), (b"elem.reduce_blocks", 0)):
        assert L.shm_get_tuning(key, C.addressof(v)) == 0 and v.value == dflt, key


def test_generator_gradient_buckets_partition_the_flat_buffer():
    """The data-parallel bucket plan (G gradients in reverse-layer buckets) covers every element of the flat
    gradient exactly once, and every bucket's trigger layer is the lowest layer stored in it."""
    import torch
    from shmgan_amd.model import Arena, Generator
    dev = torch.device("cpu")
    for F in (16, 64):
        g = Generator(64, F, dev, Arena(dev), lambda n: None)
        plan = g.grad_buckets()
        seen = torch.zeros(g.P.n, dtype=torch.int32)
        for trig, slices in plan:
            for lo, hi in slices:
                assert 0 <= lo < hi <= g.P.n
                seen[lo:hi] += 1
            if trig is not None:
                assert slices[0][0] == g.P.offsets[2 * trig]
        assert int(seen.min()) == 1 and int(seen.max()) == 1
        assert [t for t, _ in plan] == [16, 10, 4, None]
        if F == 64:
            mb = [sum(hi - lo for lo, hi in sl) * 4 / 1e6 for _, sl in plan]
            assert abs(sum(mb) - 74.1) < 0.1 and mb[1] > 45


def test_fused_in_bwd_scratch_size_and_timeout_report():
    """Host side of the one-pass bf16 InstanceNorm backward (include/shmgan_hip.h: SHM_IN_BWD_FUSED_DOUBLES): ops.in_bwd_fused_doubles is the
    header's macro, the scratch exists only for bfloat16 activations, and a set timeout word is reported by name (Trainer.losses raises on it)."""
    import re
    import torch
    from shmgan_amd import ops
    from shmgan_amd.model import Arena, _fused_scratch
    hdr = (Path(__file__).resolve().parent.parent / "include" / "shmgan_hip.h").read_text()
    m = re.search(r"#define SHM_IN_BWD_FUSED_DOUBLES\(batch, hw, c\)\s*\\\n((?:.*\\\n)*.*)", hdr)
    assert m, "macro not found"
    expr = re.sub(r"\(size_t\)", "", m.group(1)).replace("\\\n", " ").replace("/", "//").replace("SHM_IN_BWD_FUSED_CB(c)", "min(c, 64)")
    for batch, hw, c in ((40, 65536, 64), (8, 16384, 128), (3, 1024, 512), (1, 4096, 8)):
        assert ops.in_bwd_fused_doubles(batch, hw, c) == eval(expr, {"batch": batch, "hw": hw, "c": c}), (batch, hw, c)
    A = Arena(torch.device("cpu"))
    assert _fused_scratch(A, torch.float32, 2, 4096, 64) is None
    t = _fused_scratch(A, torch.bfloat16, 2, 4096, 64)
    assert t.dtype == torch.float64 and t.numel() == ops.in_bwd_fused_doubles(2, 4096, 64) and float(t.abs().max()) == 0.0
    assert _fused_scratch(A, torch.bfloat16, 2, 4096, 64) is t
    u = _fused_scratch(A, torch.bfloat16, 4, 1024, 128)
    assert A.fused_timeouts() == []
    u[-1] = 4.9e-324          # the timeout word is a uint32 in the last float64: any set bit
    assert A.fused_timeouts() == ["bwd/fused/4x1024x128"]


def test_every_pipeline_barrier_waits_for_its_lds_reads():
    """Source lint for the race fixed in round 2: a raw s_barrier in a kernel source would let a wave enter the barrier with
    fragment reads still queued (the refill of the stage is issued right after it); kernels use SHM_LDS_BARRIER() -- s_waitcnt
    lgkmcnt(0) + s_barrier -- or __syncthreads()."""
    from pathlib import Path
    csrc = Path(_lib.__file__).resolve().parent / "csrc"
    for f in sorted(csrc.glob("*.hip")):
        assert "__builtin_amdgcn_s_barrier" not in f.read_text(), f.name
    common = (csrc / "common.h").read_text()
    i = common.index("#define SHM_LDS_BARRIER()")
    body = common[i:i + 400]
    assert body.index("s_waitcnt lgkmcnt(0)") < body.index("__builtin_amdgcn_s_barrier()")


def test_built_library_has_no_store_data_hazard():
    """ISA lint for the fault found in round 4 (tools/check_isa_hazards.py): no VALU write into the data registers of a 12- / 16-byte store
    within two wait states of it, anywhere in the library's gfx950 code -- hipcc does not space that out after buffer stores with an SGPR
    soffset, and an inline-asm v_max landed there.  The scanner itself is checked on a crafted listing."""
    import importlib.util
    from pathlib import Path
    root = Path(_lib.__file__).resolve().parents[1]
    spec = importlib.util.spec_from_file_location("check_isa_hazards", root / "tools" / "check_isa_hazards.py")
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    bad = "0000 <k>:\n\tbuffer_store_dwordx4 v[16:19], v104, s[36:39], s66 offen\n\tv_max_f32_e32 v17, v12, v0\n"
    ok1 = "0000 <k>:\n\tbuffer_store_dwordx4 v[16:19], v104, s[36:39], s66 offen\n\ts_nop 1\n\tv_max_f32_e32 v17, v12, v0\n"
    ok2 = "0000 <k>:\n\tglobal_store_dwordx4 v[16:17], v[20:23], off\n\tv_lshl_add_u64 v[16:17], v[16:17], 0, 16\n\tv_mov_b32_e32 v1, v2\n\tv_mov_b32_e32 v20, v2\n"
    bad2 = "0000 <k>:\n\tglobal_store_dwordx4 v[16:17], v[20:23], off\n\tv_mov_b32_e32 v1, v2\n\tv_pk_add_f32 v[22:23], v[2:3], v[4:5]\n"
    assert len(chk.scan(bad)) == 1 and len(chk.scan(bad2)) == 1 and not chk.scan(ok1) and not chk.scan(ok2)
    # ... and its second rule: no VALU read of an MFMA's VGPR result within passes + 2 wait states of it, per MFMA shape (another MFMA may
    # chain on it and counts as its own passes; a VALU write to the register supersedes the result; an unconditional branch ends the window)
    bad3 = "0000 <k>:\n\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]\n\tv_mov_b32_e32 v7, v8\n\tv_max_f32_e32 v9, v1, v2\n"
    bad4 = "0000 <k>:\n\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]\n\ts_nop 4\n\tv_max_f32_e32 v9, v1, v2\n"          # 8 passes want 10
    bad5 = "0000 <k>:\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\ts_nop 3\n\tv_cvt_pk_bf16_f32 v9, v0, v1\n"   # 4 passes want 6
    ok3 = "0000 <k>:\n\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]\n\ts_nop 7\n\ts_nop 1\n\tv_max_f32_e32 v9, v1, v2\n"
    ok4 = "0000 <k>:\n\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]\n\tv_mfma_f32_16x16x4_f32 v[0:3], v6, v7, v[0:3]\n\tv_mfma_f32_16x16x4_f32 a[0:3], v6, v7, a[0:3]\n"
    ok5 = "0000 <k>:\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[4:7], v[8:11], v[0:3]\n\ts_nop 5\n\tv_cvt_pk_bf16_f32 v9, v0, v1\n"
    ok6 = "0000 <k>:\n\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]\n\tv_mov_b32_e32 v1, 0\n\tv_max_f32_e32 v9, v1, v1\n"                 # v1 rewritten
    ok7 = "0000 <k>:\n\tv_mfma_f32_16x16x4_f32 v[0:3], v4, v5, v[0:3]\n\ts_branch 12\n\tv_max_f32_e32 v9, v1, v2\n"                       # not fall-through
    assert len(chk.scan(bad3)) == 1 and len(chk.scan(bad4)) == 1 and len(chk.scan(bad5)) == 1
    assert not chk.scan(ok3) and not chk.scan(ok4) and not chk.scan(ok5) and not chk.scan(ok6) and not chk.scan(ok7)
    assert chk.main(["check_isa_hazards.py", str(_lib.LIB_PATH)]) == 0


def test_bench_refuses_a_rank_count_that_disagrees_with_the_launcher():
    """bench.py --gpus N is the contract the driver computes scaling from: with WORLD_SIZE != N it must not print a line at all
    (round 2 printed n_gpus: 1 for `--gpus 8` launched as plain python).  The check runs before anything touches the GPU."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == "" and "WORLD_SIZE=1" in r.stderr
