"""A barrier of the one-pass bf16 InstanceNorm backward that gives up must be fatal in the step that tripped it (round 6; VERDICT r5 item 3,
advisor r5 "medium").  include/shmgan_hip.h: shm_set_abort_words, shm_adam_clip; the step is SHM.py:859-872 (gradients -> clip -> Adam).

  * "elem.fused_test_stall" = 1 makes every barrier of in_bwd_fused8_kernel wait for a block that does not exist: the launch times out, sets the
    scratch's word, the device abort word and the pinned host word; shm_adam_clip then leaves w, m, v bit for bit;
  * a trainer whose step tripped keeps the weights of the last good step, raises KernelAbortError at the next train_step / losses() /
    checkpoint, and works again after clear_abort();
  * the launcher takes the one-pass form only when twice a group's blocks fit the device;
  * stress: 200 full-size bf16 steps with the weight-gradient lane on and every gradient bucket going through RCCL on the side stream
    (one-rank group, SHM_DP_FORCE=1: an RCCL kernel co-resident with the barrier kernels) -- no timeout, the kernel's outputs equal the two passes' to 2e-3 and the
    step's gradients to the bf16 tolerance per tensor, and the kernel on fixed operands gives the same bits on a third stream while steps are running.
"""
import multiprocessing as mp
import os
import socket

import numpy as np
import pytest
import torch

from oracle import step_torch as st

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops():
    from shmgan_amd import ops
    return ops


@pytest.fixture(autouse=True)
def _reset():
    yield
    _ops().set_tuning("reset", 0)
    _ops().set_abort_words(None, None)


def _words():
    dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    host = torch.zeros(1, dtype=torch.int32).pin_memory()
    return dev, host, host.numpy()


def _fused_call(n=2, h=32, c=64, stall=0):
    ops = _ops()
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.standard_normal((n, h, h, c)).astype(np.float32)).cuda().to(BF)
    g = torch.from_numpy(rng.standard_normal((n, h, h, c)).astype(np.float32)).cuda().to(BF)
    stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    dz = torch.zeros_like(a)
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    ops.set_tuning("elem.fused_test_stall", stall)
    ops.in_bwd(g, c, None, 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=scr)
    kern = ops.last_kernel()
    ops.set_tuning("elem.fused_test_stall", 0)
    return scr, dz, kern


def test_timeout_sets_the_words_and_adam_refuses():
    ops = _ops()
    dev, host, hnp = _words()
    ops.set_abort_words(dev, host)
    n = 4096
    rng = np.random.default_rng(0)
    w, m, v, g = (torch.from_numpy(rng.standard_normal(n).astype(np.float32)).cuda() for _ in range(4))
    v = v.abs()
    # armed and clean: the update is applied
    w0 = w.clone()
    ops.adam_clip(w, m, v, g, n, 1e-3, 0.5, 0.99, 1e-7, 1.0)
    torch.cuda.synchronize()
    assert not torch.equal(w, w0) and int(hnp[0]) == 0 and int(dev.item()) == 0
    # a clean fused call leaves the words alone
    scr, _, kern = _fused_call(stall=0)
    torch.cuda.synchronize()
    assert kern.startswith("in_bwd_fused8_kernel") and int(hnp[0]) == 0 and int(dev.item()) == 0
    assert int(scr[-1:].view(torch.int64).item()) == 0
    # every barrier times out: all three words are set -- the host word without any device-to-host copy
    scr, dz, kern = _fused_call(stall=1)
    torch.cuda.synchronize()
    assert kern.startswith("in_bwd_fused8_kernel")
    assert int(hnp[0]) == 1, "the pinned host word is written by the kernel itself"
    assert int(dev.item()) != 0 and int(scr[-1:].view(torch.int64).item()) != 0
    assert bool(torch.isfinite(dz.float()).all())                      # the launch completed (wrong means, no hang)
    # the scratch is clean again apart from its timeout word (counters, flags, means)
    rows = (2 * (32 * 32 * 64 // 16384) * 3 * 64 + 1) // 2
    assert int(scr[rows:-1].view(torch.int64).abs().sum().item()) == 0
    # the optimizer kernel applies nothing while the device word is set
    w1, m1, v1 = w.clone(), m.clone(), v.clone()
    ops.adam_clip(w, m, v, g, n, 1e-3, 0.5, 0.99, 1e-7, 1.0)
    torch.cuda.synchronize()
    assert torch.equal(w, w1) and torch.equal(m, m1) and torch.equal(v, v1)
    dev.zero_()
    ops.adam_clip(w, m, v, g, n, 1e-3, 0.5, 0.99, 1e-7, 1.0)
    torch.cuda.synchronize()
    assert not torch.equal(w, w1)
    # disarmed: a timeout is only recorded in the scratch's own word
    ops.set_abort_words(None, None)
    hnp[0] = 0
    scr, _, _ = _fused_call(stall=1)
    torch.cuda.synchronize()
    assert int(hnp[0]) == 0 and int(dev.item()) == 0 and int(scr[-1:].view(torch.int64).item()) != 0


def test_trainer_keeps_the_last_good_weights_and_raises():
    from shmgan_amd import KernelAbortError, ShmGANwithSSpecSeg
    ops = _ops()
    S, F, B = 64, 32, 1
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype="bfloat16").build()
    inp, sf = st.make_inputs(B, S), st.style_factor_intended(S)
    m.train_step(*inp, draws=st.make_draws(0, B, S, F), style_factor=sf)
    torch.cuda.synchronize()
    assert m.arena.fused_timeouts(clear=False) == []
    assert any(k[0].startswith("bwd/fused/") for k in m.arena.t if isinstance(k[0], str)), "the one-pass form is not in this step"
    wG, wD, mG = m.G.P.flat.clone(), m.D.P.flat.clone(), m.G.P.m.clone()
    ops.set_tuning("elem.fused_test_stall", 1)
    m.train_step(*inp, draws=st.make_draws(1, B, S, F), style_factor=sf)          # trips; the host cannot know yet
    ops.set_tuning("elem.fused_test_stall", 0)
    torch.cuda.synchronize()
    assert torch.equal(m.G.P.flat, wG) and torch.equal(m.D.P.flat, wD) and torch.equal(m.G.P.m, mG), "Adam ran behind a tripped barrier"
    with pytest.raises(KernelAbortError):
        m.train_step(*inp, draws=st.make_draws(2, B, S, F), style_factor=sf)
    with pytest.raises(KernelAbortError):
        m.losses()
    with pytest.raises(KernelAbortError):
        m._save_checkpoint()
    assert torch.equal(m.G.P.flat, wG)
    m.clear_abort()
    m.train_step(*inp, draws=st.make_draws(2, B, S, F), style_factor=sf)
    torch.cuda.synchronize()
    assert not torch.equal(m.G.P.flat, wG) and np.isfinite(m.losses()["total_Generator_loss"])
    m.release()


def test_release_reports_an_unseen_abort():
    from shmgan_amd import KernelAbortError, ShmGANwithSSpecSeg
    ops = _ops()
    S, F, B = 64, 32, 1
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype="bfloat16").build()
    inp, sf = st.make_inputs(B, S), st.style_factor_intended(S)
    ops.set_tuning("elem.fused_test_stall", 1)
    m.train_step(*inp, draws=st.make_draws(0, B, S, F), style_factor=sf)
    ops.set_tuning("elem.fused_test_stall", 0)
    with pytest.raises(KernelAbortError):
        m.release()
    assert m.G is None and not m.arena.t            # released all the same


def test_group_must_fit_the_device_twice():
    """256 slices per group need 512 resident blocks: an MI355X holds 1024 (768 of the pooled form).  "elem.fused_max_slices" = 512 lifts the
    knob's limit, the residency rule still refuses 512-slice groups of the pooled form (2 x 512 > 768) and takes the plain one (2 x 512 <= 1024)."""
    ops = _ops()
    rng = np.random.default_rng(1)
    ops.set_tuning("elem.fused_max_slices", 512)
    ops.set_tuning("elem.fused_hold", 1)             # the kernel that holds g and a (the g-held one halves the group: tests/test_in_bwd_fused_gpu.py)
    for pool, expect_fused in ((False, True), (True, False)):
        n, h, w, c = 1, 256, 512, 64              # 131072 pixels = 512 slices of 256
        a = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).cuda().to(BF)
        g = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).cuda().to(BF)
        g2 = torch.from_numpy(rng.standard_normal((n, h // 2, w // 2, c)).astype(np.float32)).cuda().to(BF) if pool else None
        stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
        ops.in_stats(a, c, stats, n, h * w, c, 1e-6)
        dz = torch.zeros_like(a)
        db = torch.zeros(c, dtype=torch.float64, device="cuda")
        red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.in_bwd_fused_doubles(n, h * w, c), dtype=torch.float64, device="cuda")
        ops.in_bwd(g, c, g2, c if pool else 0, a, c, stats, red, dz, c, db, n, h, w, c, 0.2, fused=scr)
        kern = ops.last_kernel()
        torch.cuda.synchronize()
        assert kern.startswith("in_bwd_fused8_kernel") == expect_fused, (pool, kern)
        assert int(scr[-1:].view(torch.int64).item()) == 0


# ------------------------------------------------------------------------------------------------------------------ stress
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stress_worker(port, q, steps):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SHM_DP_FORCE="1")
        os.environ.pop("SHM_NO_WGRAD_LANE", None)
        import torch.distributed as dist
        torch.cuda.set_device(0)
        from shmgan_amd import ShmGANwithSSpecSeg, ops
        from shmgan_amd.dist import exchange_active, init_process_group
        init_process_group("nccl", device=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl" and exchange_active()
        S, F, B = 256, 64, 8
        m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype="bfloat16").build()
        assert m._get_lane().stream is not None
        inp, sf = st.make_inputs(B, S), st.style_factor_intended(S)
        dr = st.make_draws(0, B, S, F)

        def grads(fused):
            ops.set_tuning("elem.fused_bwd", fused)
            m.train_step(*inp, draws=dr, style_factor=sf, apply=False)
            torch.cuda.synchronize()
            return torch.cat([m.G.P.grad, m.D.P.grad]).clone()
        g1 = grads(1)
        kinds = sorted({k[0] for k in m.arena.t if isinstance(k[0], str) and k[0].startswith("bwd/fused/")})
        g2 = grads(1)
        g0 = grads(0)

        def worst_rel(x, y):
            worst, off = 0.0, 0
            for M in (m.G, m.D):
                for o, z in zip(M.P.offsets, M.P._sizes):
                    a, b = x[off + o:off + o + z].double(), y[off + o:off + o + z].double()
                    den = float(b.norm())
                    if den > 0:
                        worst = max(worst, float((a - b).norm()) / den)
                off += M.P.n
            return worst
        repeat, worst = worst_rel(g1, g2), worst_rel(g1, g0)
        ops.set_tuning("elem.fused_bwd", 1)
        m._reducer.probe = []
        for step in range(steps):
            m.train_step(*inp, style_factor=sf, next_batch=inp)
        m._check_abort(sync=True)
        comm = m._reducer.comm_summary(steps)
        tripped = m.arena.fused_timeouts(clear=False)
        loss = m.losses()["total_Generator_loss"]
        # the one-pass kernel on fixed operands (a mean-25 gradient: one missed partial row moves dz by ~10 %) on a third stream WHILE steps
        # run on the main stream, the weight-gradient lane and the RCCL stream: every repeat must give the bits of the idle-GPU result
        rng = np.random.default_rng(11)
        n, h, c = 8, 128, 128
        a = torch.from_numpy((rng.standard_normal((n, h, h, c)) * 1.5 + 0.3).astype(np.float32)).cuda().to(BF)
        g = torch.from_numpy((rng.standard_normal((n, h, h, c)) + 25.0).astype(np.float32)).cuda().to(BF)
        stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
        ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
        red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")

        def one(fused=True):
            dz = torch.zeros_like(a)
            db = torch.zeros(c, dtype=torch.float64, device="cuda")
            ops.in_bwd(g, c, None, 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=scr)
            assert ops.last_kernel().startswith("in_bwd_fused8_kernel") == fused
            return dz, db
        ref = one()
        ops.set_tuning("elem.fused_bwd", 0)
        two = one(False)
        ops.set_tuning("elem.fused_bwd", 1)
        torch.cuda.synchronize()
        op_rel = max(float((ref[0].double() - two[0].double()).norm() / two[0].double().norm()),
                     float((ref[1] - two[1]).norm() / two[1].norm()))
        side = torch.cuda.Stream()
        outs = []
        for _ in range(6):
            m.train_step(*inp, style_factor=sf, next_batch=inp)
            with torch.cuda.stream(side):
                outs += [one() for _ in range(8)]
        torch.cuda.synchronize()
        m._check_abort(sync=True)
        bitwise = all(torch.equal(dz, ref[0]) and torch.equal(db, ref[1]) for dz, db in outs)
        q.put(("ok", bitwise, repeat, (worst, op_rel), tripped, float(loss), comm, kinds, int(m._abort_np[0])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:       # the parent asserts on the message
        import traceback
        q.put(("raised", repr(e), traceback.format_exc()))


def test_stress_200_steps_beside_rccl_and_the_weight_gradient_lane():
    steps = 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_stress_worker, args=(_free_port(), q, steps))
    p.start()
    r = q.get(timeout=900)
    p.join(timeout=120)
    assert r[0] == "ok", r
    _, bitwise, repeat, worst, tripped, loss, comm, kinds, host_word = r
    assert p.exitcode == 0
    assert len(kinds) >= 4, kinds                              # the one-pass form is what ran
    assert tripped == [] and host_word == 0
    assert comm["collectives"] == 6, comm                      # every bucket went through RCCL beside the barrier kernels
    assert bitwise, "the one-pass InstanceNorm backward is not bitwise repeatable under load"
    assert repeat <= 1e-6, repeat                              # whole step run to run: the order of the float64 statistics atomics only
    # fused against two passes: the kernel's own outputs to the bound of tests/test_in_bwd_fused_gpu.py (bf16 rounding of dz); the whole step's
    # gradients per tensor to the bf16 path's tolerance (a dz that rounds the other way is a 2^-9 relative change which the layers below inherit)
    assert worst[1] <= 2e-3 and worst[0] <= 2e-2, worst
    assert np.isfinite(loss)
