"""ctypes binding of libshmgan_hip.so (the C ABI declared in include/shmgan_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails
this module raises.  `build()` (re)compiles the library in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# SHM_LIB_PATH: load another build of the same ABI (tools/ablate_conv.py times ablated kernels this way)
LIB_PATH = Path(os.environ["SHM_LIB_PATH"]) if os.environ.get("SHM_LIB_PATH") else _HERE / "libshmgan_hip.so"
CSRC = _HERE / "csrc"
HEADER = _HERE.parent / "include" / "shmgan_hip.h"
SOURCES = ["conv_igemm.hip", "conv_wreg16.hip", "conv_pingpong.hip", "conv_wgrad.hip", "conv_wgrad_x3.hip", "conv_fwd_x3.hip", "conv_rgb.hip", "norm_elem.hip", "color.hip", "imgloss.hip", "specseg.hip", "data.hip"]
F32, BF16 = 0, 1                 # SHM_F32 / SHM_BF16 of include/shmgan_hip.h
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-munsafe-fp-atomics",
               "-Wall", "-Wno-unused-function", "-Wno-unused-local-typedef"]

# per-source extra flags.  conv_pingpong.hip: its epilogue runs beside the partner wave's MFMAs, where the packed f32 instructions the SLP vectoriser
# forms (v_pk_add_f32 / v_pk_mul_f32) cost several times their scalar pairs (MI355X guide, cycle constants)
EXTRA_FLAGS = {"conv_pingpong.hip": ["-fno-slp-vectorize"]}

P, I, Z, F = C.c_void_p, C.c_int, C.c_size_t, C.c_float

# name -> (restype, argtypes); must mirror include/shmgan_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "shm_version": (I, []),
    "shm_last_error": (C.c_char_p, []),
    "shm_last_kernel": (C.c_char_p, []),
    "shm_set_tuning": (I, [C.c_char_p, I]),
    "shm_get_tuning": (I, [C.c_char_p, P]),
    "shm_transpose_taps": (I, [P, P, I, I, I, I, I, P]),
    "shm_transpose_taps_multi": (I, [I, P, P, P, P, P, P, I, P]),
    "shm_cast_f32": (I, [P, P, Z, I, P]),
    "shm_conv2d_fwd": (I, [P, P, I, I, I, P, P, P, I, I, I, I, I, I, I, I, F, I, P]),
    "shm_conv2d_in_fwd": (I, [P, P, I, I, I, P, P, P, I, I, I, I, I, I, I, I, F, P, P, F, I, P]),
    "shm_conv2d_dgrad": (I, [P, I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "shm_conv2d_transpose_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, F, I, P]),
    "shm_conv2d_wgrad_workspace": (Z, [I, I, I, I, I, I]),
    "shm_conv2d_wgrad_partial": (I, [P, P, I, I, I, P, I, I, I, I, I, I, I, I, I, P, Z, I, P, P]),
    "shm_conv2d_wgrad_reduce": (I, [P, P, Z, I, I, P]),
    "shm_conv2d_wgrad": (I, [P, P, I, I, I, P, I, P, I, I, I, I, I, I, I, I, I, P, Z, I, P]),
    "shm_in_stats": (I, [P, I, P, I, I, I, F, I, P]),
    "shm_in_apply": (I, [P, I, P, P, P, I, I, I, I, I, P]),
    "shm_in_apply_pool": (I, [P, I, P, P, P, I, P, I, I, I, I, I, I, P]),
    "shm_in_norm_table": (I, [P, P, P, I, I, P]),
    "shm_conv2d_in_fwd_norm": (I, [P, P, I, I, I, P, P, I, P, P, P, I, I, I, I, I, I, I, I, F, P, P, F, P, P, I, P]),
    "shm_conv2d_norm_prepare": (I, [P, P, P, I, I, P, P, I, I, I, I, I, P]),
    "shm_conv2d_norm_supported": (I, [I, I, I, I, I, I, I, I, I, I]),
    "shm_conv2d_wgrad_norm": (I, [P, P, I, I, I, P, P, I, P, I, P, I, I, I, I, I, I, I, I, I, P, Z, I, P]),
    "shm_conv2d_wgrad_partial_norm": (I, [P, P, I, I, I, P, P, I, P, I, I, I, I, I, I, I, I, I, P, Z, I, P, P]),
    "shm_conv2d_wgrad_norm_workspace": (Z, [I, I, I, I, I, I, I]),
    "shm_conv2d_wgrad_norm_finish": (I, [P, P, P, I, I, I, I, I, I, P]),
    "shm_in_bwd_keep_dz_sums": (I, [P]),
    "shm_in_bwd_fused_scratch": (I, [P, Z]),
    "shm_set_abort_words": (I, [P, P]),
    "shm_set_clock_probe": (I, [P]),
    "shm_conv2d_wgrad_norm_supported": (I, [I, I, I, I, I, I, I, I, I, I, I]),
    "shm_in_pool": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "shm_in_bwd": (I, [P, I, P, I, P, I, P, P, P, I, P, I, I, I, I, F, I, P]),
    "shm_conv2d_dgrad_gsum": (I, [P, I, P, P, P, I, I, I, I, I, I, I, I, I, I, P, I, P, P, I, P, I, P]),
    "shm_conv2d_fwd_gsum": (I, [P, P, I, I, I, P, P, P, I, I, I, I, I, I, I, I, F, P, I, P, I, P]),
    "shm_in_bwd_apply": (I, [P, I, P, I, P, I, P, P, P, P, P, P, I, P, I, I, I, I, F, I, P]),
    "shm_sum_input_channels": (I, [P, I, I, C.c_uint, P, P]),
    "shm_conv3x3_dgrad_sum1": (I, [P, I, P, P, I, I, I, I, I, I, I, I, P]),
    "shm_lrelu_bwd": (I, [P, I, P, I, P, I, P, P, Z, I, F, I, P]),
    "shm_avgpool2_fwd": (I, [P, I, P, I, I, I, I, I, I, P]),
    "shm_cvt_f64_f32": (I, [P, P, Z, I, P]),
    "shm_zero": (I, [P, Z, P]),
    "shm_head_fwd": (I, [P, I, P, P, P, Z, I, F, I, P]),
    "shm_head_bwd": (I, [P, I, P, P, P, P, I, P, P, P, Z, I, F, I, P]),
    "shm_head_in_fwd": (I, [P, I, P, P, P, P, P, I, I, I, F, I, P]),
    "shm_head_in_bwd": (I, [P, I, P, P, P, P, P, P, I, P, P, P, P, I, I, I, F, I, P]),
    "shm_in_bwd_rank1": (I, [P, P, P, I, P, P, P, I, P, I, I, I, I, F, I, P]),
    "shm_patch_fwd": (I, [P, I, P, P, I, I, I, I, F, I, P]),
    "shm_patch_bwd": (I, [P, I, P, P, P, P, P, I, P, I, I, I, I, F, I, P]),
    "shm_dense_fwd": (I, [P, P, P, I, I, I, I, P]),
    "shm_dense_bwd": (I, [P, P, P, P, P, I, I, I, I, P]),
    "shm_mul_mask": (I, [P, P, P, Z, F, I, P]),
    "shm_rgb2yuv_std": (I, [P, P, P, P, I, Z, P]),
    "shm_avg_cbcr": (I, [P, P, P, P, P, P, Z, P]),
    "shm_build_gen_input": (I, [P, P, P, P, P, P, I, I, P, I, I, Z, I, P]),
    "shm_cyc_input_bwd": (I, [P, I, I, P, I, Z, I, P]),
    "shm_yuv2rgb": (I, [P, P, P, P, P, I, I, I, Z, I, P]),
    "shm_pack_rgb16": (I, [P, P, P, I, Z, I, P]),
    "shm_rgb16_to_dy": (I, [P, I, P, Z, I, I, P]),
    "shm_randn": (I, [P, Z, F, C.c_ulonglong, C.c_uint, P]),
    "shm_keep_mask": (I, [P, Z, F, C.c_ulonglong, C.c_uint, P]),
    "shm_dhead_losses": (I, [P, P, P, P, P, P, I, I, F, I, P]),
    "shm_image_losses_workspace": (Z, [I, I]),
    "shm_image_losses": (I, [P, P, P, P, P, P, I, F, P, P, P, P, Z, I, I, P]),
    "shm_pack_channels": (I, [P, I, I, I, P, I, Z, P]),
    "shm_bn_apply": (I, [P, I, P, P, P, P, F, P, I, Z, I, P]),
    "shm_maxpool2_fwd": (I, [P, I, P, I, I, I, I, I, P]),
    "shm_conv2d_transpose2x2_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, F, I, P]),
    "shm_head_sigmoid_fwd": (I, [P, I, P, P, P, Z, I, P]),
    "shm_spec_loss": (I, [P, P, P, P, P, I, Z, P]),
    "shm_mask_pool_pack": (I, [P, P, I, I, I, I, I, P]),
    "shm_add_bcast": (I, [P, P, P, I, Z, I, I, I, P]),
    "shm_sum_groups": (I, [P, P, I, Z, I, I, I, I, P]),
    "shm_resize_bilinear_u8": (I, [P, I, I, I, P, I, I, F, I, P]),
    "shm_adam_clip": (I, [P, P, P, P, Z, F, F, F, F, F, P]),
}

def header_functions():
    """Names declared in include/shmgan_hip.h (used by the ABI test)."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.findall(r"\b(shm_[a-z0-9_]+)\s*\(", txt)


def build(force=False, verbose=False, jobs=None):
    """Compile csrc/*.hip into libshmgan_hip.so with hipcc (cross-compiles without a GPU): one object per source under
    csrc/_obj/ (rebuilt when older than its source or a shared header, or when the compile command changed; up to `jobs` at a
    time), then one link.  Objects and the library are written under temporary names and renamed into place, so that
    several ranks building a stale tree at once never read each other's half-written files."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    srcs = [CSRC / s for s in SOURCES]
    shared = [CSRC / "common.h", CSRC / "ablate.h", CSRC / "tapgemm.h", CSRC / "wgrad.h", CSRC / "x3split.h", HEADER]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    # what the objects were built WITH: compiler, flags and source list (a change of HIPCC_FLAGS must not relink stale objects)
    stamp_txt = hashlib.sha256(" ".join([hipcc, *cflags, *SOURCES, repr(sorted(EXTRA_FLAGS.items()))]).encode()).hexdigest()
    objdir = CSRC / "_obj"
    objdir.mkdir(exist_ok=True)
    stamp = objdir / "flags.sha256"
    same_cmd = stamp.exists() and stamp.read_text().strip() == stamp_txt
    deps = srcs + shared
    if not force and same_cmd and LIB_PATH.exists():
        newest = max(p.stat().st_mtime for p in deps)
        if LIB_PATH.stat().st_mtime >= newest:
            return LIB_PATH
    hdr_time = max(p.stat().st_mtime for p in shared)
    todo = []
    for src in srcs:
        obj = objdir / (src.stem + ".o")
        if force or not same_cmd or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_time):
            todo.append((obj, [hipcc, *cflags, *EXTRA_FLAGS.get(src.name, []), "-c", str(src)]))

    def run(cmd, out):
        tmp = out.with_name(f".{out.name}.{os.getpid()}.tmp")
        if verbose:
            print(" ".join(cmd + ["-o", str(out)]), flush=True)
        try:
            subprocess.run(cmd + ["-o", str(tmp)], check=True)
            os.replace(tmp, out)
        finally:
            if tmp.exists():
                tmp.unlink()
    with ThreadPoolExecutor(max_workers=jobs or min(4, os.cpu_count() or 1)) as ex:
        list(ex.map(lambda t: run(t[1], t[0]), todo))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *[str(objdir / (src.stem + ".o")) for src in srcs]], LIB_PATH)
    stamp.write_text(stamp_txt + "\n")
    return LIB_PATH


_lib = None


def lib():
    """Load the shared library (once).  Raises if it is missing: there is no fallback."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  shmgan_amd has no CPU/PyTorch fallback.")
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class ShmError(RuntimeError):
    pass


def check(rc, name):
    if rc != 0:
        msg = lib().shm_last_error()
        raise ShmError(f"{name} failed ({rc}): {msg.decode() if msg else ''}")
