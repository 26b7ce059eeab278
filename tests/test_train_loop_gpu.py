"""The callers of train_step on the device: `train(args)` (SHM.py:889-1139, called by main.py:107), the dataset
loader's stream safety, and the run-to-run reproducibility of a step."""
import argparse

import numpy as np
import pytest
import torch

from oracle import data_np as dn
from oracle import step_torch as st
from util import host, rel_l2

pytestmark = pytest.mark.gpu


def _write_dataset(root, n, rng, hw=(40, 50)):
    from PIL import Image
    from shmgan_amd.data import PSD_SUBDIRS
    ref = {}
    for v, sub in enumerate(PSD_SUBDIRS):
        (root / sub).mkdir(parents=True)
        for i in range(n):
            img = rng.integers(0, 256, (hw[0], hw[1], 3)).astype(np.uint8)
            Image.fromarray(img).save(root / sub / f"img_{i:03d}.png")
            ref[(v, i)] = img
    return ref


def test_train_entry_runs_epochs_and_checkpoints(tmp_path):
    """main.py:103-107: ShmGANwithSSpecSeg(args).train(args) on 4 five-view tuples: (4 - 1) steps per epoch (SHM.py:979),
    TARGET_LABELS redrawn per step (SHM.py:986), a checkpoint per `checkpoint_save_step` epochs + one at exit, newest 3
    kept; a second train() restores the latest one (SHM.py:949-951)."""
    from shmgan_amd import ShmGANwithSSpecSeg
    _write_dataset(tmp_path / "data", 4, np.random.default_rng(1))
    args = argparse.Namespace(mode="train", image_size=32, batch_size=1, filter_size=16, num_epochs=2, g_lr=2e-5, d_lr=2e-5,
                              beta1=0.5, beta2=0.99, data_dir=str(tmp_path / "data"), checkpoint_save_dir=str(tmp_path / "ckpt"),
                              log_dir=str(tmp_path / "logs"), log_step=1, checkpoint_save_step=1)
    shmgan = ShmGANwithSSpecSeg(args)
    lines, labels = [], []
    step0 = shmgan.train_step

    def spy(*a, **k):
        labels.append(shmgan.TARGET_LABELS)
        return step0(*a, **k)

    shmgan.train_step = spy
    n = shmgan.train(args, print_fn=lines.append)
    torch.cuda.synchronize()
    assert n == 6 and shmgan.G.P.iterations == 6 and shmgan.D.P.iterations == 6 and shmgan.epoch == 1
    assert len(set(labels)) == 6 and all(0.8 <= t <= 1.2 for t in labels)
    assert np.isfinite(shmgan.losses()["total_Generator_loss"])
    names = sorted(p.name for p in (tmp_path / "ckpt").glob("*.npz"))
    assert names == ["ckpt-1.npz", "ckpt-2.npz", "ckpt-3.npz"]
    assert (tmp_path / "logs" / "Generator_summary.txt").read_text().count("Conv2D") >= 23
    assert any("Start of Training Epoch 1" in l for l in lines)
    # resume
    args.num_epochs = 1
    again = ShmGANwithSSpecSeg(args)
    lines2 = []
    assert again.train(args, max_steps=2, print_fn=lines2.append) == 2
    assert any("Latest checkpoint restored" in l for l in lines2)
    assert again.G.P.iterations == 8
    assert sorted(p.name for p in (tmp_path / "ckpt").glob("*.npz")) == ["ckpt-3.npz", "ckpt-4.npz", "ckpt-5.npz"]


def test_checkpoints_are_atomic_resumable_and_mode_checked(tmp_path):
    """Round-2 advisor findings on the checkpoint path: the file is written under a temporary name and renamed (no
    half-written ckpt-N.npz), a corrupt newest checkpoint falls back to the previous one, a resumed run continues the
    draw streams (flags generator state + Philox counter) instead of replaying the first run's opening steps, and a
    checkpoint of the other attention mode is refused with a clear error."""
    from shmgan_amd import ShmGANwithSSpecSeg
    S, F, B = 32, 16, 1
    kw = dict(image_size=S, filter_size=F, batch_size=B, checkpoint_save_dir=str(tmp_path))
    m = ShmGANwithSSpecSeg(**kw).build()
    inp = st.make_inputs(B, S)
    for _ in range(3):
        m.train_step(*inp)                       # default draws: advances _rng and _draw_count
    p1 = m._save_checkpoint()
    m.train_step(*inp)
    p2 = m._save_checkpoint()
    assert sorted(q.name for q in tmp_path.iterdir()) == ["ckpt-1.npz", "ckpt-2.npz"]         # no .tmp left behind
    state2 = (m._draw_count, m._rng.bit_generator.state, m.G.P.iterations)
    nxt = m._default_draws(B).flags                                                            # what step 5 would draw
    with open(p2, "r+b") as f:                   # a kill mid-write as older versions could leave it
        f.truncate(1000)
    r = ShmGANwithSSpecSeg(**kw)
    assert r._restore_latest() == p1 and r._draw_count == 3 and r.G.P.iterations == 3
    m._save_checkpoint()                         # ckpt-3 (state after the extra _default_draws call above)
    r2 = ShmGANwithSSpecSeg(**kw)
    assert r2._restore_latest().endswith("ckpt-3.npz")
    assert r2._draw_count == state2[0] + 1 and r2.G.P.iterations == state2[2]
    r3 = ShmGANwithSSpecSeg(**kw)
    r3.load_npz(p1)
    for _ in range(1):
        r3.train_step(*inp)
    assert r3._draw_count == state2[0] and r3._rng.bit_generator.state == state2[1]            # same stream position as m had
    assert r3._default_draws(B).flags == nxt
    with pytest.raises(KeyError, match="attention"):
        ShmGANwithSSpecSeg(attention="live", **kw).load_npz(p1)


def test_lookahead_prologue_changes_nothing_but_the_issue_order():
    """train_step(next_batch=): the next step's weight-independent prologue (rgb->yuv, standardise, CbCr average, SpecSeg
    mask) is issued at the end of the current step into the other buffer slot and picked up by the next call.  Three
    steps with the look-ahead (tensors, then a callable) must leave the same weights, losses and masks as three steps
    without it, to the run-to-run bound of the step (float64 atomics in the statistics sums: 1e-6, see
    test_step_is_reproducible_run_to_run); a look-ahead for OTHER tensors than the next call's is discarded."""
    from shmgan_amd import ShmGANwithSSpecSeg
    S, F, B = 64, 16, 2
    batches = [st.make_inputs(B, S, rank=r) for r in range(3)]
    dev = [[torch.from_numpy(a).cuda() for a in b] for b in batches]
    runs = []
    for mode in ("plain", "ahead", "wrong"):
        m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
        out = []
        for i in range(3):
            nb = None
            if mode == "ahead" and i < 2:
                nb = dev[i + 1] if i == 0 else (lambda i=i: dev[i + 1])
            if mode == "wrong":
                nb = dev[i]                     # not the batch the next call gets
            m.train_step(*dev[i], draws=st.make_draws(i, B, S, F), next_batch=nb)
            if mode == "ahead" and i < 2:
                assert m._prefetched is not None and m._prefetched.slot == (i + 1) % 2
            out.append((dict(m.losses()), m.specular_candidate.clone()))
        torch.cuda.synchronize()
        runs.append((out, m.G.P.flat.clone(), m.D.P.flat.clone()))
    for other in runs[1:]:
        for (l0, k0), (l1, k1) in zip(runs[0][0], other[0]):
            for k, v in l0.items():
                if k != "ssim":
                    assert abs(l1[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, v, l1[k])
            assert rel_l2(host(k1), host(k0)) <= 1e-6
        assert rel_l2(host(other[1]), host(runs[0][1])) <= 1e-6 and rel_l2(host(other[2]), host(runs[0][2])) <= 1e-6


def test_loader_batches_survive_an_asynchronous_consumer(tmp_path):
    """The consumer stream runs far behind the host (as train_step does when nobody reads the losses): every batch is
    copied by a kernel queued behind a long matmul chain and dropped at once.  The loader allocates its outputs on its
    own stream and records the consumer stream on them, so the next batches' resize kernels must not land in memory the
    queued copies still read."""
    from shmgan_amd.data import PolarDataset
    S, n = 64, 8
    ref = _write_dataset(tmp_path, n, np.random.default_rng(2), hw=(70, 90))
    ds = PolarDataset(str(tmp_path), S, batch_size=1)
    big = torch.randn(4096, 4096, device="cuda")
    copies = []
    for batch in ds:
        acc = big
        for _ in range(6):                       # ~50 ms of queued work per batch in front of the copies
            acc = acc @ big * 1e-3
        copies.append([v.clone() for v in batch])
        del batch, acc
    torch.cuda.synchronize()
    assert len(copies) == n
    for i, c in enumerate(copies):
        for v in range(5):
            assert np.abs(host(c[v][0]) - dn.load_view(ref[(v, i)], S, True)).max() < 2e-6, (i, v)


def test_step_is_reproducible_run_to_run():
    """The only floating-point atomics in the step are float64 adds (InstanceNorm statistics slots, IN-backward sums, bias
    gradients); weight gradients use fixed-order slab sums.  Stated bound: two identical steps agree in every named loss
    to 1e-6 relative and in both flat gradients to rel-L2 <= 1e-6 (float64 rounding of a differently ordered sum is
    ~1e-16 and survives the cast to fp32 only if it crosses a rounding boundary)."""
    from shmgan_amd import ShmGANwithSSpecSeg
    S, F, B = 128, 32, 2
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
    inp, dr = st.make_inputs(B, S), st.make_draws(4, B, S, F)
    runs = []
    for _ in range(2):
        m.train_step(*inp, draws=dr, apply=False)
        torch.cuda.synchronize()
        runs.append((dict(m.losses()), m.G.P.grad.clone(), m.D.P.grad.clone(), m.gen_Y.clone()))
    (l0, g0, d0, y0), (l1, g1, d1, y1) = runs
    for k, v in l0.items():
        if k != "ssim":
            assert abs(l1[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, v, l1[k])
    assert rel_l2(host(y1), host(y0)) <= 1e-6
    assert rel_l2(host(g1), host(g0)) <= 1e-6 and rel_l2(host(d1), host(d0)) <= 1e-6


@pytest.mark.parametrize("dt", ["bfloat16", "float32"])
def test_full_size_step_is_reproducible_over_many_runs(dt):
    """BASELINE configs[1] geometry (S=256, F=64, B=8), ten repetitions of the same step: every repetition agrees with the first
    in gen_Y, the discriminator outputs, the named losses and the flat gradients to the float64-atomics bound (1e-6).
    This is the race detector of the LDS-DMA pipelines: a fragment read overtaken by the refill of its stage (a barrier entered
    with ds_reads still queued -- see SHM_LDS_BARRIER in csrc/common.h) showed up here as one stale 16-byte weight chunk in about
    one launch in thirty of the bf16 128-wide halo block, i.e. in roughly every third step."""
    from shmgan_amd import ShmGANwithSSpecSeg
    S, F, B = 256, 64, 8
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype=dt).build()
    inp, dr = st.make_inputs(B, S), st.make_draws(5, B, S, F)
    ref = None
    for r in range(10):
        m.train_step(*inp, draws=dr, apply=False)
        torch.cuda.synchronize()
        cur = (dict(m.losses()), m.gen_Y.clone(), m.D.ctx["rf"].clone(), m.D.ctx["cls"].clone(), m.G.P.grad.clone(), m.D.P.grad.clone())
        if ref is None:
            ref = cur
            continue
        # (in practice bitwise equal; the bound leaves room for a float64 statistics sum that crosses an fp32 rounding boundary
        # when its atomics arrive in another order -- the race moved these outputs by 1e-2)
        for i in (1, 2, 3):
            assert rel_l2(host(cur[i]), host(ref[i])) <= 1e-6, (r, i)
        for k, v in ref[0].items():
            if k != "ssim":
                assert abs(cur[0][k] - v) <= 1e-6 * max(1.0, abs(v)), (r, k, v, cur[0][k])
        assert rel_l2(host(cur[4]), host(ref[4])) <= 1e-6 and rel_l2(host(cur[5]), host(ref[5])) <= 1e-6, r


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_conv_entry_points_are_bitwise_reproducible(dt):
    """The hot tap-GEMM and weight-gradient kernels on a full-size layer (256 -> 256 at 64 x 64, n = 40), thirty launches each:
    identical bits every time (their only atomics are the float64 statistics, which do not feed the output tensor)."""
    from shmgan_amd import ops
    adt = torch.bfloat16 if dt == "bf16" else torch.float32
    n, h, c = 40, 64, 256
    torch.manual_seed(1)
    x = torch.randn((n, h, h, c), device="cuda").to(adt)
    dy = torch.randn((n, h, h, c), device="cuda").to(adt)
    w = torch.randn((3, 3, c, c), device="cuda") * 0.05
    wk = torch.zeros(9 * c * c, device="cuda", dtype=adt)
    ops.transpose_taps(w, wk, 9, c, c, c)
    wop = w.to(adt)
    b = torch.randn(c, device="cuda")
    stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h, h, c, c, 3) // 4 + 1024, device="cuda")
    dy2 = torch.randn((n, h // 2, h // 2, c), device="cuda").to(adt)            # stride-2 weight gradient of the same x (round 4: bf16 = the
    ref = None                                                                  # eight-wave halo form with even / odd column runs)
    ops.set_tuning("wgrad.bf16_wide", 4)                                        # ... and the eight-wave form at unit stride as well
    for r in range(30):
        y = torch.empty((n, h, h, c), device="cuda", dtype=adt)
        dx = torch.empty((n, h, h, c), device="cuda", dtype=adt)
        dw = torch.empty((3, 3, c, c), device="cuda")
        ops.conv2d_in_fwd(x, None, 0, c, 0, wk, b, y, c, n, h, h, c, c, 3, 1, 0.2, stats, 1e-6, scratch=scr)
        ops.conv2d_dgrad(dy, c, wop, dx, None, c, c, 0, n, h, h, c, c, 3, 1)
        ops.conv2d_wgrad(x, None, 0, c, 0, dy, c, dw, n, h, h, c, c, c, 3, 1, 0, ws)
        k1 = ops.last_kernel()
        dw2 = torch.empty((3, 3, c, c), device="cuda")
        ops.conv2d_wgrad(x, None, 0, c, 0, dy2, c, dw2, n, h, h, c, c, c, 3, 2, 0, ws)
        k2 = ops.last_kernel()
        torch.cuda.synchronize()
        if ref is None:
            ref = (y, dx, dw, dw2)
            assert (k1, k2) == (("wgrad_halo8_bf16_kernel<0>", "wgrad_halo8_bf16_kernel<1>") if dt == "bf16" else ("wgrad_halo_kernel", "wgrad_halo_kernel<0, true>")), (k1, k2)
        else:
            assert torch.equal(y, ref[0]) and torch.equal(dx, ref[1]) and torch.equal(dw, ref[2]) and torch.equal(dw2, ref[3]), r
    ops.set_tuning("reset", 0)
