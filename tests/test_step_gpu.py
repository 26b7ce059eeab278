"""GPU parity of the loss kernels, whole generator / discriminator passes and the whole
train_step against the float64 oracle (SURVEY 8(c) tolerances)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import specseg_torch as sp
from oracle import step_torch as st
from util import check_grad_fixture as _check_grad_fixture, pin_kinks as _pin_kinks, cosine, dev, host, rel_l2, t64

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["executed", "intended"])
def test_dhead_losses(mode):
    """Both class-logit gradient modes of shm_dhead_losses against the oracle's softmax_xent: "executed" = TF's fused
    kernel (softmax - labels; differs from the derivative on D1's label row [0,0,0,0,T], T != 1), "intended" = autograd."""
    from shmgan_amd import ops
    rng = np.random.default_rng(20)
    B, npatch, T = 3, 4, 1.07
    rf = rng.standard_normal((12 * B, npatch))
    cls = rng.standard_normal((12 * B, 5))
    rft, clst = t64(rf).requires_grad_(True), t64(cls).requires_grad_(True)
    sl = lambda g, k=0: slice((g + k) * B, (g + k + 1) * B)       # group start in units of B
    mse = lambda a, t: ((a - t) ** 2).mean(dim=1)
    def xent(lg, k, w=1.0):
        lab = torch.zeros_like(lg)
        lab[:, k] = w
        return st.softmax_xent(lg, lab, mode)
    D1, D2 = sl(0), sl(6)
    D3 = [sl(1, k) for k in range(5)]
    D4 = [sl(7, k) for k in range(5)]
    D1_RF, D3_RF = mse(rft[D1], T), sum(mse(rft[s], T) for s in D3)
    D2_RF = mse(rft[D2], T) + (rft[D1] ** 2).mean(dim=1)
    D4_RF = sum(mse(rft[D4[k]], T) + (rft[D3[k]] ** 2).mean(dim=1) for k in range(5)) + D2_RF
    D1_c, D3_c = xent(clst[D1], 4, T), sum(xent(clst[D3[k]], k) for k in range(5))
    D4_c = sum(xent(clst[D4[k]], k) for k in range(5))
    tot_d = ((D1_c + D3_c) / 6 + (D2_RF + D4_RF) / 6 + 0.5 * D4_c + 10 * D4_c).mean()
    tot_g = ((D1_RF + D3_RF) / 6).mean()
    gd_rf, gd_cls = torch.autograd.grad(tot_d, [rft, clst], retain_graph=True)
    gg_rf, = torch.autograd.grad(tot_g, [rft])
    loss = torch.empty(16, dtype=torch.float64, device="cuda")
    drf_d = torch.empty((12 * B, npatch), device="cuda")
    dcls_d = torch.empty((12 * B, 5), device="cuda")
    drf_g = torch.empty((6 * B, npatch), device="cuda")
    ops.dhead_losses(dev(rf), dev(cls), loss, drf_d, dcls_d, drf_g, B, npatch, T,
                     ops.XENT_TF_FUSED if mode == "executed" else ops.XENT_INTENDED)
    assert rel_l2(host(drf_d), gd_rf.numpy()) < 1e-5
    assert rel_l2(host(dcls_d), gd_cls.numpy()) < 1e-5
    # the two modes differ on the D1 rows by (T - 1) * softmax / (6 B) and nowhere else
    other = torch.empty_like(dcls_d)
    ops.dhead_losses(dev(rf), dev(cls), loss, drf_d, other, drf_g, B, npatch, T,
                     ops.XENT_INTENDED if mode == "executed" else ops.XENT_TF_FUSED)
    diff = host(other) - host(dcls_d)
    sign = 1.0 if mode == "executed" else -1.0
    want = sign * (T - 1.0) * torch.softmax(t64(cls[:B]), -1).numpy() / (6.0 * B)
    assert np.abs(diff[:B] - want).max() < 1e-6 and np.abs(diff[B:]).max() == 0.0
    with pytest.raises(RuntimeError):
        ops.dhead_losses(dev(rf), dev(cls), loss, drf_d, other, drf_g, B, npatch, T, 7)
    assert rel_l2(host(drf_g), gg_rf.numpy()[:6 * B]) < 1e-5
    L = host(loss)
    ref = [D1_RF.sum(), D3_RF.sum(), (rft[D1] ** 2).mean(dim=1).sum(),
           sum((rft[s] ** 2).mean(dim=1) for s in D3).sum(), mse(rft[D2], T).sum(),
           sum(mse(rft[s], T) for s in D4).sum(), D1_c.sum(), D3_c.sum(), D4_c.sum()]
    assert rel_l2(L[:9], [float(r.detach()) for r in ref]) < 1e-5


@pytest.mark.parametrize("B,S,flags", [(1, 32, (False, True, False, False, False)), (2, 48, (True, False, False, True, False))])
def test_image_losses(B, S, flags):
    from shmgan_amd import ops
    rng = np.random.default_rng(21)
    npix = S * S
    orig = [rng.random((B, S, S, 3)) for _ in range(5)]
    ds = [st.per_image_standardization(st.rgb_to_yuv(t64(o)))[0] for o in orig]
    cbcr = sum(d[..., 1:] for d in ds) / 5.0
    gen_y = t64(rng.standard_normal((B, S, S, 1)) * 0.5 + 1.0).requires_grad_(True)
    cyc_y = t64(rng.standard_normal((5 * B, S, S, 1)) * 0.5 + 1.0).requires_grad_(True)
    gen_rgb = st.yuv_to_rgb(torch.cat([gen_y, cbcr], 3))
    cyuv = [torch.cat([cyc_y[k * B:(k + 1) * B], cbcr], 3) for k in range(5)]
    crgb = [st.yuv_to_rgb(c) for c in cyuv]
    l1 = lambda a, b: (a - b).abs().mean(dim=(1, 2, 3))
    L1 = (sum(l1(crgb[k], t64(orig[k])) for k in range(4)) + l1(gen_rgb, t64(orig[4]))) / 5 + 10 * l1(crgb[4], t64(orig[4]))
    ssims = [st.ssim(st.rescale_01(cyuv[k]), st.rescale_01(ds[k])) for k in range(5)]
    sl = [torch.zeros(B, dtype=torch.float64) if flags[k] else -torch.log((1 + ssims[k]) / 2) for k in range(5)]
    ssim_loss = (sl[0] + sl[1] + sl[2] + sl[3] + 10 * sl[4]) / 5
    sf = 3.0e-3          # large enough that the style term is visible in the gradient
    content = ((cyuv[4] - ds[0]) ** 2).mean(dim=(1, 2, 3))
    style = sf * ((st.gram_matrix(cyuv[4]) - st.gram_matrix(ds[4])) ** 2).mean(dim=(1, 2))
    tot = (10 * L1 + 10 * ssim_loss + 10 * (100 * style + content)).mean()
    rg, rc = torch.autograd.grad(tot, [gen_y, cyc_y])

    fmask = sum(1 << k for k in range(5) if flags[k])
    od = [dev(o) for o in orig]
    dd = [dev(d.numpy()) for d in ds]
    loss = torch.empty(32, dtype=torch.float64, device="cuda")
    dg = torch.empty((B, S, S, 1), device="cuda")
    dc = torch.empty((5 * B, S, S, 1), device="cuda")
    ws = torch.empty(ops.image_losses_workspace(B, S) // 4 + 1, device="cuda")
    optr = (C.c_void_p * 5)(*[t.data_ptr() for t in od])
    dptr = (C.c_void_p * 5)(*[t.data_ptr() for t in dd])
    ops.image_losses(dev(gen_rgb.detach().numpy()), dev(torch.cat(crgb, 0).detach().numpy()), dev(cyc_y.detach().numpy()),
                     dev(cbcr.numpy()), optr, dptr, fmask, sf, loss, dg, dc, ws, B, S)
    L = host(loss)
    assert abs(L[0] - float(l1(gen_rgb, t64(orig[4])).sum())) < 1e-5 * B
    for k in range(5):
        assert abs(L[1 + k] - float(l1(crgb[k], t64(orig[k])).sum())) < 1e-5 * B
        assert abs(L[6 + k] - float(ssims[k].sum())) < 2e-5 * B, (k, L[6 + k], float(ssims[k].sum()))
        assert abs(L[11 + k] - float(sl[k].sum())) < 2e-5 * B
    assert abs(L[16] - float(content.sum())) < 1e-5 * B * max(1.0, float(content.max()))
    assert abs(L[17] - float(style.sum())) < 1e-5 * max(1.0, float(style.sum()))
    assert rel_l2(host(dg), rg.numpy()) < 1e-4
    assert rel_l2(host(dc), rc.numpy()) < 1e-4, cosine(host(dc), rc.numpy())


def _mk(S, F, B):
    from shmgan_amd import ShmGANwithSSpecSeg
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
    g, d, gb, db = st.init_params(F, S)
    # the product's own init must equal the oracle's (same seeds, same order)
    for a, b in zip(m.G.get_weights(), g):
        assert np.array_equal(a, b)
    for a, b in zip(m.D.get_weights(), d):
        assert np.array_equal(a, b)
    return m, (g, d, gb, db)


def test_generator_forward_backward():
    S, F, B = 32, 16, 2
    m, (g, d, gb, db) = _mk(S, F, B)
    rng = np.random.default_rng(30)
    x = rng.standard_normal((B, S, S, 10))
    dy = rng.standard_normal((B, S, S, 1))
    gv = [t64(a).requires_grad_(True) for a in g]
    xt = t64(x).requires_grad_(True)
    yt = st.generator_forward(gv, [t64(b) for b in gb], xt, F)
    grads = torch.autograd.grad(yt, gv + [xt], t64(dy))
    x16 = torch.zeros((B, S, S, 16), device="cuda")
    x16[..., :10] = dev(x)
    y = m.G.forward(x16, "t")
    assert np.abs(host(y) - yt.detach().numpy()).max() < 1e-4
    m.G.zero_grad()
    dx = m.G.backward(dev(dy), "t", need_dx=True)
    m.G.finish_grads()
    torch.cuda.synchronize()
    assert rel_l2(host(dx)[..., :10], grads[-1].numpy()) < 1e-3
    for i, (got, ref) in enumerate(zip(m.G.P.grads, grads[:-1])):
        r = rel_l2(host(got), ref.numpy())
        assert r < 1e-3 and cosine(host(got), ref.numpy()) > 0.9999, (i, r)


def test_discriminator_forward_backward():
    S, F, B = 64, 16, 3
    m, (g, d, gb, db) = _mk(S, F, B)
    rng = np.random.default_rng(31)
    x = rng.random((B, S, S, 3))
    s = S // 32
    noise = rng.standard_normal((B, S, S, 3)) * 0.1
    keep = (rng.random((B, s, s, 16 * F)) >= 0.2).astype(np.float32)
    dv = [t64(a).requires_grad_(True) for a in d]
    xt = t64(x).requires_grad_(True)
    rf, cls = st.discriminator_forward(dv, [t64(b) for b in db], xt, t64(noise), t64(keep))
    g_rf, g_cls = rng.standard_normal(tuple(rf.shape)), rng.standard_normal(tuple(cls.shape))
    grads = torch.autograd.grad([rf, cls], dv + [xt], [t64(g_rf), t64(g_cls)], retain_graph=True)
    gx_only, = torch.autograd.grad([rf], [xt], [t64(g_rf)])
    rfd, clsd = m.D(dev(x), training=True, noise=dev(noise), keep_mask=dev(keep))
    assert np.abs(host(rfd) - rf.detach().numpy()).max() < 1e-4
    assert np.abs(host(clsd) - cls.detach().numpy()).max() < 1e-4
    m.D.zero_grad()
    m.D.backward_params(dev(g_rf), dev(g_cls))
    torch.cuda.synchronize()          # weight gradients run on the wgrad lane (second stream)
    for i, (got, ref) in enumerate(zip(m.D.P.grads, grads[:-1])):
        r = rel_l2(host(got), ref.numpy())
        assert r < 1e-3 and cosine(host(got), ref.numpy()) > 0.9999, (i, r)
    dx = m.D.backward_input(B, dev(g_rf))
    assert rel_l2(host(dx)[..., :3], gx_only.numpy()) < 1e-3


@pytest.mark.parametrize("S,F,B,step", [(64, 16, 1, 0), (64, 16, 2, 1), (64, 32, 1, 0)])      # F = 32: K = 32 layers (narrow weights-in-registers forms)
def test_train_step_parity(S, F, B, step):
    m, (g, d, gb, db) = _mk(S, F, B)
    inp = st.make_inputs(B, S)
    dr = st.make_draws(step, B, S, F)
    sf = st.style_factor_intended(S)
    sw = sp.init_specseg(seed=44 + step)
    m.SpecSeg.set_weights(sw)
    m.train_step(*inp, draws=dr, style_factor=sf, apply=False)
    torch.cuda.synchronize()
    # The oracle evaluates every LeakyReLU on the side of the kink the device took (see
    # oracle.step_torch._act): at |z| ~ 1e-7 float32 rounding can flip the sign, the derivative
    # jumps 5x there, and a single such element moved a whole layer's gradient by 1e-2 in
    # rel-L2 (DESIGN.md "parity").  Values are unaffected (<= 1e-6).
    masks = {"g1": m.G.lrelu_masks("g1"), "cyc": m.G.lrelu_masks("cyc"), "d": m.D.lrelu_masks()}
    ref = st.train_step(g, d, gb, db, inp, dr, sf, F, masks=masks)
    free = st.train_step(g, d, gb, db, inp, dr, sf, F, need_grads=False, specseg=sw)   # un-pinned oracle: values
    assert np.abs(host(m.specular_candidate) - free["outs"]["specular_candidate"].numpy()).max() < 1e-5
    got = m.losses()
    for k, v in free["losses"].items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    for k, v in ref["losses"].items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    assert np.abs(host(m.gen_Y) - ref["outs"]["gen_Y"].numpy()).max() < 1e-4
    assert np.abs(host(m.gen_rgb) - ref["outs"]["gen_rgb"].numpy()).max() < 1e-4
    for name, P, rg in (("D", m.D.P, ref["gD"]), ("G", m.G.P, ref["gG"])):
        for i, (got_g, r) in enumerate(zip(P.grads, rg)):
            r = r.numpy()
            if np.linalg.norm(r) < 1e-12:
                continue
            e = rel_l2(host(got_g), r)
            assert e < 1e-3 and cosine(host(got_g), r) > 0.9999, (name, i, e)
    # optimizer: apply and compare with the oracle's clip+Adam
    gv = [t64(a).clone() for a in g]
    dvv = [t64(a).clone() for a in d]
    sg = st.AdamState([torch.zeros_like(a) for a in gv], [torch.zeros_like(a) for a in gv])
    sd = st.AdamState([torch.zeros_like(a) for a in dvv], [torch.zeros_like(a) for a in dvv])
    # (a) the kernel: the oracle's clip+Adam applied to the DEVICE's gradients reproduces the device's weights to fp32 rounding
    dev_gd = [t64(host(t)) for t in m.D.P.grads]
    dev_gg = [t64(host(t)) for t in m.G.P.grads]
    st.adam_apply(dvv, dev_gd, sd, 2e-5, 0.5, 0.99)
    st.adam_apply(gv, dev_gg, sg, 2e-5, 0.5, 0.99)
    # (b) end to end: the oracle's clip+Adam on the ORACLE's gradients.  The first Adam step is lr * g / (|g| + 1e-6): where a
    # gradient element is itself ~1e-6 the 1e-3 relative gradient error can flip its sign, so single weights may differ by
    # up to 2 lr; all but a vanishing fraction must agree to a tenth of a step
    gv2, dvv2 = [t64(a).clone() for a in g], [t64(a).clone() for a in d]
    sg2 = st.AdamState([torch.zeros_like(a) for a in gv2], [torch.zeros_like(a) for a in gv2])
    sd2 = st.AdamState([torch.zeros_like(a) for a in dvv2], [torch.zeros_like(a) for a in dvv2])
    st.adam_apply(dvv2, ref["gD"], sd2, 2e-5, 0.5, 0.99)
    st.adam_apply(gv2, ref["gG"], sg2, 2e-5, 0.5, 0.99)
    m.optimizer_D.apply(m.D.P)
    m.optimizer_G.apply(m.G.P)
    torch.cuda.synchronize()
    for got_w, r, r2 in zip(m.G.P.vars + m.D.P.vars, gv + dvv, gv2 + dvv2):
        assert np.abs(host(got_w) - r.numpy()).max() < 1e-8 + 2e-7 * float(r.abs().max())
        e = np.abs(host(got_w) - r2.numpy())
        assert e.max() <= 2.02 * 2e-5 and (e > 2e-6).mean() < 1e-2, (e.max(), (e > 2e-6).mean())


def test_train_step_properties_full_size():
    """S=256, F=64, B=1: size-independent checks (no oracle at this size in the GPU suite):
    finite losses, IN outputs normalised, and gradient linearity in the batch rule (B=2 of the same
    sample twice == B=1)."""
    from shmgan_amd import ShmGANwithSSpecSeg
    S, F = 256, 64
    inp1 = st.make_inputs(1, S)
    dr1 = st.make_draws(3, 1, S, F)
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=1).build()
    m.train_step(*inp1, draws=dr1, apply=False)
    torch.cuda.synchronize()
    l1 = dict(m.losses())
    g1 = m.G.P.grad.clone()
    d1 = m.D.P.grad.clone()
    assert all(np.isfinite(v) for k, v in l1.items() if k != "ssim")
    assert m.G.count_params() == 18525569 and m.D.count_params() == 6605504
    # the same sample twice -> identical mean loss and gradients
    inp2 = [np.concatenate([a, a], 0) for a in inp1]
    dr2 = st.StepDraws(dr1.flags, dr1.target_label, np.concatenate([dr1.noise[:1], dr1.noise[:1], dr1.noise[1:], dr1.noise[1:]], 0),
                       np.concatenate([dr1.keep_mask[:1], dr1.keep_mask[:1], dr1.keep_mask[1:], dr1.keep_mask[1:]], 0))
    m2 = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=2).build()
    m2.train_step(*inp2, draws=dr2, apply=False)
    torch.cuda.synchronize()
    l2 = m2.losses()
    # The two runs dispatch different tap-GEMM variants (the tile choice depends on the grid size: at B = 1 the deep layers
    # take the 64 x 64 DMA tile), whose K orders differ in the last bit; a handful of pre-activations then sit on the other
    # side of a LeakyReLU kink (see test_train_step_parity), which moves a loss by ~1e-5 (bound: the 1e-4 of SURVEY 8(c)'s
    # named-loss tolerance) and the gradient by up to ~1e-3 in rel-L2.
    for k in l1:
        if k != "ssim":
            assert abs(l1[k] - l2[k]) <= 1e-4 * max(1.0, abs(l1[k])), (k, l1[k], l2[k])
    assert rel_l2(host(m2.G.P.grad), host(g1)) < 3e-3 and cosine(host(m2.G.P.grad), host(g1)) > 0.99999
    assert rel_l2(host(m2.D.P.grad), host(d1)) < 3e-3 and cosine(host(m2.D.P.grad), host(d1)) > 0.99999


@pytest.mark.parametrize("name", ["step_S64_F16_B1.npz", "step_S64_F16_B2.npz"])
def test_golden_fixture_on_device(name):
    """HIP path against the committed golden vectors (tests/golden, made by oracle/make_golden.py):
    named losses, gen_Y, SSIM values, gradient norms and random projections."""
    from pathlib import Path
    gold = np.load(Path(__file__).resolve().parent / "golden" / name)
    S, F, B, step = [int(v) for v in gold["meta"]]
    m, _ = _mk(S, F, B)
    m.SpecSeg.set_weights(sp.init_specseg(seed=44 + step))
    m.train_step(*st.make_inputs(B, S), draws=st.make_draws(step, B, S, F), style_factor=st.style_factor_intended(S), apply=False)
    torch.cuda.synchronize()
    got = m.losses()
    assert np.abs(host(m.specular_candidate) - gold["specular_candidate"]).max() < 1e-5
    for k in got:
        if k != "ssim":
            v = float(gold[f"loss/{k}"])
            assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    assert np.abs(host(m.gen_Y) - gold["gen_Y"]).max() < 1e-4
    assert np.abs(np.array(got["ssim"]) - gold["ssim"].mean(axis=1)).max() < 1e-4
    _check_grad_fixture(m, gold)


# f32_split 1: "wgrad.f32_split" (the fp32 weight gradients from six bf16 MFMA products), 2: plus "conv.f32_split" (the forward and input-gradient
# convolutions too, bench.py --dtype f32x3) -- same bounds
@pytest.mark.parametrize("f32_split", [0, 1, 2])
@pytest.mark.parametrize("name", ["step_S256_F64_B1.npz", "step_S256_F64_B8.npz"])
def test_golden_fixture_full_size(name, f32_split):
    """BASELINE configs[1] at full size (S=256, F=64; B=8 is the bench batch) against the committed float64-oracle
    fixture: every named loss, gen_Y (subsampled values + per-sample moments), SSIM, the SpecSeg mask, and per-tensor
    gradient norms and projections of all 53 weight tensors.  This is the only place the fp32 kernels' full-size
    dispatch (halo 128, DMA 128x128 / 128x64 / 64x128, halo weight gradient) is compared with the oracle end to end.
    The fixture carries the float64 step's near-kink pre-activations; the device is pinned to their signs (_pin_kinks),
    so ALL 53 tensors are held to 1e-3 at B=1 and at the bench batch B=8."""
    from pathlib import Path
    gold = np.load(Path(__file__).resolve().parent / "golden" / name)
    S, F, B, step, sub = [int(v) for v in gold["meta"]]
    from shmgan_amd import ops
    m, _ = _mk(S, F, B)
    m.SpecSeg.set_weights(sp.init_specseg(seed=44 + step))
    m.before_backward, pinned = _pin_kinks(m, gold)
    try:
        ops.set_tuning("wgrad.f32_split", 1 if f32_split else 0)
        ops.set_tuning("conv.f32_split", 1 if f32_split == 2 else 0)
        timer = ops.KernelTimer() if f32_split else None
        ops.TIMER = timer
        m.train_step(*st.make_inputs(B, S), draws=st.make_draws(step, B, S, F), style_factor=st.style_factor_intended(S), apply=False)
        torch.cuda.synchronize()
    finally:
        ops.TIMER = None
        ops.set_tuning("reset", 0)
    if f32_split:                                        # the knob did take the step's 3x3 unit-stride weight gradients
        names = list(timer.summary())
        assert any(k.startswith("wgrad_halo_x3_kernel") for k in names) and not any(k == "wgrad_halo_kernel" for k in names), names
        assert any(k.startswith("tapgemm_halo_x3_kernel") for k in names) == (f32_split == 2), names
    got = m.losses()
    for k in got:
        if k != "ssim":
            v = float(gold[f"loss/{k}"])
            assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    gy = host(m.gen_Y)
    assert np.abs(gy[:, ::sub, ::sub] - gold["gen_Y_sub"]).max() < 1e-4
    assert np.abs(gy.reshape(B, -1).sum(1) / gold["gen_Y_sum"] - 1).max() < 1e-5
    assert np.abs((gy.reshape(B, -1) ** 2).sum(1) / gold["gen_Y_sq"] - 1).max() < 1e-5
    assert np.abs(host(m.specular_candidate)[:, ::sub, ::sub] - gold["specular_candidate_sub"]).max() < 1e-5
    assert np.abs(np.array(got["ssim"]) - gold["ssim"].mean(axis=1)).max() < 1e-4
    # un-pinned, 4 M pre-activations per 64-channel layer put a few elements of EVERY layer on the other side of zero and the
    # typical tensor sits at ~1.5e-3 (round 2: median held to 5e-3, worst to 5e-2); pinned, every tensor meets the contract
    listed, flipped = sum(v[0] for v in pinned.values()), sum(v[1] for v in pinned.values())
    print(f"kink pins: {flipped} of {listed} listed near-kink elements had the other sign on the device")
    assert listed > 0 and flipped < 0.2 * listed
    _check_grad_fixture(m, gold, med_tol=1e-3, worst_tol=1e-3)


def test_train_step_parity_full_size():
    """The whole step at BASELINE's layer widths and image size (S=256, F=64, B=1) against the float64 oracle evaluated
    on the GPU box's host cores (~1 min, ~12 GB), with the device's LeakyReLU sign pattern pinned as in
    test_train_step_parity: named losses 1e-4, gen_Y 1e-4, every weight-gradient tensor rel-L2 <= 1e-3 and cosine >= 0.9999."""
    S, F, B, step = 256, 64, 1, 0
    m, (g, d, gb, db) = _mk(S, F, B)
    inp = st.make_inputs(B, S)
    dr = st.make_draws(step, B, S, F)
    sf = st.style_factor_intended(S)
    m.train_step(*inp, draws=dr, style_factor=sf, apply=False)
    torch.cuda.synchronize()
    masks = {"g1": m.G.lrelu_masks("g1"), "cyc": m.G.lrelu_masks("cyc"), "d": m.D.lrelu_masks()}
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    ref = st.train_step(g, d, gb, db, inp, dr, sf, F, masks=masks)
    got = m.losses()
    for k, v in ref["losses"].items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    assert np.abs(host(m.gen_Y) - ref["outs"]["gen_Y"].numpy()).max() < 1e-4
    worst = 0.0
    for name, P, rg in (("D", m.D.P, ref["gD"]), ("G", m.G.P, ref["gG"])):
        for i, (got_g, r) in enumerate(zip(P.grads, rg)):
            r = r.numpy()
            if np.linalg.norm(r) < 1e-12:
                continue
            e = rel_l2(host(got_g), r)
            worst = max(worst, e)
            assert e < 1e-3 and cosine(host(got_g), r) > 0.9999, (name, i, e)
    print("full-size parity: worst per-tensor rel-L2", worst)


def test_inference_path_matches_oracle(tmp_path):
    """test.py:218-297: forward-only G x6, plus the .npz weight round trip (N3)."""
    S, F, B = 64, 16, 2
    m, (g, d, gb, db) = _mk(S, F, B)
    rng = np.random.default_rng(40)
    rgb = rng.random((B, S, S, 3))
    ref = st.infer(g, gb, rgb, F)
    gen_rgb, cyc = m.infer(rgb)
    torch.cuda.synchronize()
    assert np.abs(host(gen_rgb) - ref["gen_rgb"].numpy()).max() < 1e-4
    for k in range(5):
        assert np.abs(host(cyc[k]) - ref["cyc_rgb"][k].numpy()).max() < 1e-4
    # weights / optimizer state interchange
    m.train_step(*st.make_inputs(B, S), draws=st.make_draws(0, B, S, F))
    torch.cuda.synchronize()
    p = tmp_path / "ckpt.npz"
    m.SpecSeg.set_weights(sp.init_specseg(seed=9))
    m.save_npz(p)
    from shmgan_amd import ShmGANwithSSpecSeg
    m2 = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build(seed=1, beta_seed=2)
    m2.load_npz(p)
    assert torch.equal(m2.G.P.flat, m.G.P.flat) and torch.equal(m2.D.P.flat, m.D.P.flat)
    assert torch.equal(m2.SpecSeg.flat, m.SpecSeg.flat)
    assert torch.equal(m2.G.P.m, m.G.P.m) and m2.G.P.iterations == 1
    a, _ = m.infer(rgb)
    a = host(a).copy()
    mask_a = host(m.specular_candidate).copy()           # test.py:221: the mask of the input image
    b, _ = m2.infer(rgb)
    assert np.array_equal(a, host(b)) and np.array_equal(mask_a, host(m2.specular_candidate))


def test_three_steps_track_the_oracle():
    """Three consecutive train_steps (clip + Adam applied, transposed weight copies refreshed):
    named losses of every step and the final weights follow the float64 oracle."""
    S, F, B = 64, 16, 1
    m, (g, d, gb, db) = _mk(S, F, B)
    gv = [t64(a).clone() for a in g]
    dv = [t64(a).clone() for a in d]
    sg = st.AdamState([torch.zeros_like(a) for a in gv], [torch.zeros_like(a) for a in gv])
    sd = st.AdamState([torch.zeros_like(a) for a in dv], [torch.zeros_like(a) for a in dv])
    sf = st.style_factor_intended(S)
    for step in range(3):
        inp = st.make_inputs(B, S, rank=step)
        dr = st.make_draws(step, B, S, F)
        m.train_step(*inp, draws=dr, style_factor=sf)
        torch.cuda.synchronize()
        masks = {"g1": m.G.lrelu_masks("g1"), "cyc": m.G.lrelu_masks("cyc"), "d": m.D.lrelu_masks()}
        ref = st.train_step(gv, dv, gb, db, inp, dr, sf, F, masks=masks)
        got = m.losses()
        for k, v in ref["losses"].items():
            assert abs(got[k] - v) <= 2e-4 * max(1.0, abs(v)), (step, k, got[k], v)
        st.adam_apply(dv, ref["gD"], sd, 2e-5, 0.5, 0.99)
        st.adam_apply(gv, ref["gG"], sg, 2e-5, 0.5, 0.99)
    assert m.G.P.iterations == 3 and m.D.P.iterations == 3
    # Adam's first steps move every weight by ~lr*sign(g): an element whose gradient is at the fp32
    # noise floor can legitimately step the other way, so compare the update as a whole (rel-L2 of
    # w - w0) and bound the fraction of elements that disagree by more than 10 % of their movement
    for i, (got_w, r, w0) in enumerate(zip(m.G.P.vars + m.D.P.vars, gv + dv, g + d)):
        mv_ref = r.numpy() - w0
        mv_got = host(got_w) - w0.astype(np.float64)
        if np.abs(mv_ref).max() == 0:
            continue
        assert rel_l2(mv_got, mv_ref) < 0.05, (i, rel_l2(mv_got, mv_ref))
        bad = np.abs(mv_got - mv_ref) > 0.1 * np.abs(mv_ref).max()
        assert bad.mean() < 2e-3, (i, bad.mean())


def test_long_run_stays_finite():
    """40 optimizer steps at a mid size: no NaN/Inf in losses, weights or Adam state."""
    from shmgan_amd import ShmGANwithSSpecSeg
    S, F, B = 128, 32, 2
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
    inp = st.make_inputs(B, S)
    first = None
    for step in range(40):
        m.train_step(*inp, draws=st.make_draws(step, B, S, F))
        if step in (0, 39):
            torch.cuda.synchronize()
            L = m.losses()
            assert all(np.isfinite(v) for k, v in L.items() if k != "ssim"), L
            first = first or L
    for P in (m.G.P, m.D.P):
        for t in (P.flat, P.m, P.v):
            assert bool(torch.isfinite(t).all())
    # the same batch 40 times: the L1 reconstruction term must have gone down
    assert L["L1_loss_Gen"] < first["L1_loss_Gen"]


# ------------------------------------------------------------------ SpecSeg (S1) + Spec_loss (L5)
@pytest.mark.parametrize("S,B", [(32, 2), (64, 1), (256, 1)])
def test_specseg_forward_matches_oracle(S, B):
    """SpecSeg.predict (SHM.py:492) against the float64 restatement; 1,942,801 parameters
    (SpecSeg_summary.txt:118)."""
    from shmgan_amd.model import Arena
    from shmgan_amd.specseg import SpecSeg
    net = SpecSeg(S, torch.device("cuda"), Arena(torch.device("cuda")))
    assert net.count_params() == 1942801
    w = sp.init_specseg(seed=3)
    net.set_weights(w)
    x = np.random.default_rng(S).standard_normal((B, S, S, 1)) * 2.0
    got = host(net.predict(x))
    ref = sp.specseg_forward(w, x).numpy()
    assert got.shape == ref.shape == (B, S, S, 1)
    assert np.abs(got - ref).max() < 1e-5 and rel_l2(got, ref) < 1e-5
    back = net.get_weights()
    assert all(np.array_equal(a, b) for a, b in zip(back, w))


def test_specseg_golden_fixture():
    from pathlib import Path
    from shmgan_amd.model import Arena
    from shmgan_amd.specseg import SpecSeg
    gold = np.load(Path(__file__).resolve().parent / "golden" / "specseg_S32.npz")
    net = SpecSeg(32, torch.device("cuda"), Arena(torch.device("cuda")))
    net.set_weights(sp.init_specseg(seed=3))
    assert np.abs(host(net.predict(gold["x"])) - gold["mask"]).max() < 1e-5


def test_specseg_default_init_and_summary():
    from shmgan_amd.model import Arena
    from shmgan_amd.specseg import SpecSeg
    net = SpecSeg(32, torch.device("cuda"), Arena(torch.device("cuda"))).init_random()
    lines = []
    net.summary(print_fn=lines.append)
    assert "Total params: 1,942,801" in lines and "Non-trainable params: 992" in lines
    m = host(net.predict(np.zeros((1, 32, 32, 1), np.float32)))
    assert np.allclose(m, 0.5)          # zero input, zero biases, BN mean 0 -> logit 0


def test_spec_loss_kernel():
    from shmgan_amd import ops
    rng = np.random.default_rng(8)
    B, S = 2, 16
    cyc_y = rng.standard_normal((5 * B, S, S, 1))
    cbcr = rng.standard_normal((B, S, S, 2))
    ds = [rng.standard_normal((B, S, S, 3)) for _ in range(5)]
    mask = rng.random((B, S, S, 1))
    dsd = [dev(a) for a in ds]
    ptr = (C.c_void_p * 5)(*[t.data_ptr() for t in dsd])
    loss = torch.zeros(5, dtype=torch.float64, device="cuda")
    ops.spec_loss(dev(cyc_y), dev(cbcr), ptr, dev(mask), loss, B, S * S)
    cyc = [t64(np.concatenate([cyc_y[k * B:(k + 1) * B], cbcr], axis=3)) for k in range(5)]
    total, terms = sp.spec_loss(cyc, [t64(a) for a in ds], t64(mask))
    got = loss.cpu().numpy() / (B * S * S * 3)
    assert np.allclose(got, [float(t) for t in terms], rtol=1e-5)


# ------------------------------------------------------------------ dataset loader
def test_resize_kernel_and_dataset_loader(tmp_path):
    """shm_resize_bilinear_u8 against the NumPy restatement, then the loader end to end on PNG files:
    sorted zip of five directories, resize, /255, unconditional flip_up_down (datasetLoader.py:47-61)."""
    from PIL import Image
    from oracle import data_np as dn
    from shmgan_amd import ops
    from shmgan_amd.data import PSD_SUBDIRS, PolarDataset
    rng = np.random.default_rng(12)
    for hin, win, S in ((37, 53, 32), (64, 64, 64), (20, 24, 48), (300, 200, 64)):
        img = rng.integers(0, 256, (hin, win, 3)).astype(np.uint8)
        out = torch.empty((S, S, 3), device="cuda")
        for flip in (False, True):
            ops.resize_bilinear_u8(torch.from_numpy(img).cuda(), out, 1.0 / 255.0, flip)
            assert np.abs(host(out) - dn.load_view(img, S, flip)).max() < 2e-6
    S, n = 32, 3
    ref = {}
    for v, sub in enumerate(PSD_SUBDIRS):
        (tmp_path / sub).mkdir()
        for i in range(n):
            img = rng.integers(0, 256, (40 + i, 50, 3)).astype(np.uint8)
            Image.fromarray(img).save(tmp_path / sub / f"img_{n - i:02d}.png")       # names sort in reverse of creation
            ref[(v, f"img_{n - i:02d}.png")] = img
    ds = PolarDataset(str(tmp_path), S, batch_size=1)
    assert len(ds) == n
    names = sorted(f"img_{k:02d}.png" for k in range(1, n + 1))
    for i, batch in enumerate(ds):
        assert len(batch) == 5
        for v in range(5):
            assert tuple(batch[v].shape) == (1, S, S, 3)
            assert np.abs(host(batch[v][0]) - dn.load_view(ref[(v, names[i])], S, True)).max() < 2e-6
    # and the trainer takes what the loader yields
    from shmgan_amd import ShmGANwithSSpecSeg
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=16, batch_size=1).build()
    m.train_step(*ds.batch(0))
    torch.cuda.synchronize()
    assert np.isfinite(m.losses()["total_Generator_loss"])


@pytest.mark.parametrize("S,B,dt", [(128, 2, "float32"), (128, 1, "bfloat16"), (96, 1, "float32")])
def test_other_image_sizes_run_and_track_fp32(S, B, dt):
    """image_size 128 is the reference's default (main.py:36, the committed summaries); 96 exercises maps that are not
    multiples of 16 (every 3x3 layer then takes the DMA tap GEMM, the weight gradient the generic kernel).
    Checks: finite named losses, parameter counts of the summaries at 128, and bf16 within 2e-2 of fp32."""
    from shmgan_amd import ShmGANwithSSpecSeg
    F = 64
    inp = st.make_inputs(B, S)
    dr = st.make_draws(5, B, S, F)
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype=dt).build()
    m.train_step(*inp, draws=dr, apply=False)
    torch.cuda.synchronize()
    l = m.losses()
    assert all(np.isfinite(v) for k, v in l.items() if k != "ssim")
    if S == 128:
        assert m.G.count_params() == 18525569 and m.D.count_params() == 6359744      # Generator/Discriminator_summary.txt
    assert float(m.G.P.grad.abs().sum()) > 0 and float(m.D.P.grad.abs().sum()) > 0
    if dt != "float32":
        m32 = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
        m32.train_step(*inp, draws=dr, apply=False)
        torch.cuda.synchronize()
        l32 = m32.losses()
        for k, v in l32.items():
            if k != "ssim":
                assert abs(l[k] - v) <= 2e-2 * max(1.0, abs(v)), (k, l[k], v)
