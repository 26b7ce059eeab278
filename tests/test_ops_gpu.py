"""GPU parity of every C-ABI op against the float64 CPU oracle (SURVEY 8(c) tolerances:
single conv / convT / IN op rel-L2 <= 1e-5)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import step_torch as st
from oracle import tf_ops_np as tn
from util import conv_ref, cosine, dev, host, nchw, nhwc, pad_c, rel_l2, t64

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _ops():
    from shmgan_amd import ops
    return ops


def _wk(w_hwio, cin_pad):
    """HWIO -> [t][cout][cin_pad] through the library's own transpose."""
    ops = _ops()
    k, _, cin, cout = w_hwio.shape
    wt = torch.zeros(k * k * cout * cin_pad, device="cuda")
    ops.transpose_taps(dev(w_hwio), wt, k * k, cin, cout, cin_pad)
    return wt


@pytest.mark.parametrize("n,h,cin,cout,k,s", [
    (2, 16, 16, 64, 3, 1), (1, 32, 64, 64, 3, 1), (3, 8, 32, 16, 3, 1), (2, 16, 128, 128, 3, 1),
    (2, 16, 64, 192, 3, 1), (2, 16, 64, 32, 1, 1), (2, 16, 32, 64, 3, 2), (1, 8, 256, 512, 3, 2),
    (5, 6, 16, 48, 3, 1), (1, 2, 64, 128, 3, 2),
])
def test_conv2d_fwd(n, h, cin, cout, k, s):
    ops = _ops()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, h, h, cin))
    w = rng.standard_normal((k, k, cin, cout)) * 0.1
    b = rng.standard_normal(cout)
    ref = conv_ref(x, w, s) + b
    ref = np.where(ref > 0, ref, 0.2 * ref)
    ho = ref.shape[1]
    y = torch.empty((n, ho, ho, cout), device="cuda")
    ops.conv2d_fwd(dev(x), None, 0, cin, 0, _wk(w, cin), dev(b), y, cout, n, h, h, cin, cout, k, s, 0.2)
    assert rel_l2(host(y), ref) < TOL


def test_conv2d_fwd_padded_cin_and_concat():
    ops = _ops()
    rng = np.random.default_rng(2)
    # 10 real channels in a 16-float pitch (generator input layer)
    n, h, cout = 2, 16, 64
    x = rng.standard_normal((n, h, h, 10))
    w = rng.standard_normal((3, 3, 10, cout)) * 0.1
    ref = conv_ref(x, w, 1)
    y = torch.empty((n, h, h, cout), device="cuda")
    ops.conv2d_fwd(dev(pad_c(x, 16)), None, 0, 16, 0, _wk(w, 16), None, y, cout, n, h, h, 16, cout, 3, 1, 1.0)
    assert rel_l2(host(y), ref) < TOL
    # concat of two sources (upsampled first, skip second)
    c1, c2 = 32, 48
    xa, xb = rng.standard_normal((n, h, h, c1)), rng.standard_normal((n, h, h, c2))
    w = rng.standard_normal((3, 3, c1 + c2, cout)) * 0.1
    ref = conv_ref(np.concatenate([xa, xb], -1), w, 1)
    ops.conv2d_fwd(dev(xa), dev(xb), c1, c1, c2, _wk(w, c1 + c2), None, y, cout, n, h, h, c1 + c2, cout, 3, 1, 1.0)
    assert rel_l2(host(y), ref) < TOL


@pytest.mark.parametrize("n,h,cin,cout,k,s", [
    (2, 16, 64, 64, 3, 1), (1, 8, 10, 64, 3, 1), (2, 16, 128, 64, 1, 1), (2, 16, 32, 64, 3, 2),
    (1, 32, 3, 16, 3, 2), (2, 4, 256, 512, 3, 2),
])
def test_conv2d_dgrad(n, h, cin, cout, k, s):
    ops = _ops()
    rng = np.random.default_rng(3)
    w = rng.standard_normal((k, k, cin, cout)) * 0.1
    ho = -(-h // s)
    dy = rng.standard_normal((n, ho, ho, cout))
    xt = torch.zeros(n, cin, h, h, dtype=torch.float64, requires_grad=True)
    yt = st.conv2d_same(xt, t64(w), s)
    ref, = torch.autograd.grad(yt, xt, nchw(dy))
    ref = nhwc(ref)
    ld = (cin + 15) // 16 * 16
    dx = torch.full((n, h, h, ld), 7.0, device="cuda")
    ops.conv2d_dgrad(dev(dy), cout, dev(w), dx, None, cin, ld, 0, n, h, h, cin, cout, k, s)
    assert rel_l2(host(dx)[..., :cin], ref) < TOL


def test_conv2d_dgrad_split():
    ops = _ops()
    rng = np.random.default_rng(4)
    n, h, c1, c2, cout = 2, 8, 32, 16, 64
    w = rng.standard_normal((3, 3, c1 + c2, cout)) * 0.1
    dy = rng.standard_normal((n, h, h, cout))
    xt = torch.zeros(n, c1 + c2, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(xt, t64(w), 1), xt, nchw(dy))
    ref = nhwc(ref)
    d1 = torch.empty((n, h, h, c1), device="cuda")
    d2 = torch.empty((n, h, h, c2), device="cuda")
    ops.conv2d_dgrad(dev(dy), cout, dev(w), d1, d2, c1, c1, c2, n, h, h, c1 + c2, cout, 3, 1)
    assert rel_l2(host(d1), ref[..., :c1]) < TOL
    assert rel_l2(host(d2), ref[..., c1:]) < TOL


@pytest.mark.parametrize("n,h,cin,cout", [(2, 8, 64, 64), (1, 16, 128, 64), (3, 4, 32, 16), (1, 2, 512, 512)])
def test_conv2d_transpose_fwd(n, h, cin, cout):
    ops = _ops()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, h, h, cin))
    w = rng.standard_normal((3, 3, cout, cin)) * 0.1
    b = rng.standard_normal(cout)
    ref = nhwc(st.conv2d_transpose_same(nchw(x), t64(w))) + b
    ref = np.where(ref > 0, ref, 0.2 * ref)
    y = torch.empty((n, 2 * h, 2 * h, cout), device="cuda")
    ops.conv2d_transpose_fwd(dev(x), cin, dev(w), dev(b), y, cout, n, h, h, cin, cout, 0.2)
    assert rel_l2(host(y), ref) < TOL
    # tiny case against the NumPy scatter restatement as well
    if n * h <= 12:
        ref2 = tn.conv2d_transpose_same(x, w) + b
        ref2 = np.where(ref2 > 0, ref2, 0.2 * ref2)
        assert rel_l2(host(y), ref2) < TOL


def _ws(nbytes):
    return torch.empty(max(nbytes // 4 + 1, 1024), device="cuda")


@pytest.mark.parametrize("n,h,cin,cout,k,s", [
    (2, 16, 64, 64, 3, 1), (1, 32, 16, 64, 3, 1), (3, 8, 128, 192, 3, 1), (2, 16, 64, 128, 1, 1),
    (2, 16, 32, 64, 3, 2), (4, 8, 3, 16, 3, 2), (1, 8, 10, 32, 3, 1), (7, 5, 16, 16, 3, 1),
    (3, 32, 10, 64, 3, 1), (2, 16, 10, 96, 3, 1), (2, 16, 7, 64, 3, 1),       # thin-input kernel: (tap, ci) packed into the MFMA rows
    (3, 64, 3, 64, 3, 2), (2, 32, 3, 16, 3, 2), (2, 32, 10, 32, 3, 2),        # ... and its stride-2 form (discriminator conv1)
])
def test_conv2d_wgrad(n, h, cin, cout, k, s):
    ops = _ops()
    rng = np.random.default_rng(6)
    x = rng.standard_normal((n, h, h, cin))
    ho = -(-h // s)
    dy = rng.standard_normal((n, ho, ho, cout))
    wt = torch.zeros(k, k, cin, cout, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(nchw(x), wt, s), wt, nchw(dy))
    ld = (cin + 15) // 16 * 16
    dw = torch.full((k, k, cin, cout), 3.0, device="cuda")
    ws = _ws(ops.conv2d_wgrad_workspace(n, ho, ho, cin, cout, k))
    ops.conv2d_wgrad(dev(pad_c(x, ld)), None, 0, ld, 0, dev(dy), cout, dw, n, h, h, cin, ld, cout, k, s, 0, ws)
    assert rel_l2(host(dw), ref.numpy()) < TOL
    # accumulate
    ops.conv2d_wgrad(dev(pad_c(x, ld)), None, 0, ld, 0, dev(dy), cout, dw, n, h, h, cin, ld, cout, k, s, 1, ws)
    assert rel_l2(host(dw), 2 * ref.numpy()) < TOL


def test_conv2d_wgrad_concat_and_transpose_roles():
    ops = _ops()
    rng = np.random.default_rng(7)
    n, h, c1, c2, cout = 2, 8, 32, 48, 64
    xa, xb = rng.standard_normal((n, h, h, c1)), rng.standard_normal((n, h, h, c2))
    dy = rng.standard_normal((n, h, h, cout))
    wt = torch.zeros(3, 3, c1 + c2, cout, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(nchw(np.concatenate([xa, xb], -1)), wt, 1), wt, nchw(dy))
    dw = torch.empty((3, 3, c1 + c2, cout), device="cuda")
    ws = _ws(ops.conv2d_wgrad_workspace(n, h, h, c1 + c2, cout, 3))
    ops.conv2d_wgrad(dev(xa), dev(xb), c1, c1, c2, dev(dy), cout, dw, n, h, h, c1 + c2, c1 + c2, cout, 3, 1, 0, ws)
    assert rel_l2(host(dw), ref.numpy()) < TOL
    # Conv2DTranspose weight gradient: x = dz at 2H, dy = layer input at H, stride 2
    cin_t, cout_t, hs = 32, 64, 4
    xin = rng.standard_normal((n, hs, hs, cin_t))
    dz = rng.standard_normal((n, 2 * hs, 2 * hs, cout_t))
    wt = torch.zeros(3, 3, cout_t, cin_t, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_transpose_same(nchw(xin), wt), wt, nchw(dz))
    dw = torch.empty((3, 3, cout_t, cin_t), device="cuda")
    ws = _ws(ops.conv2d_wgrad_workspace(n, hs, hs, cout_t, cin_t, 3))
    ops.conv2d_wgrad(dev(dz), None, 0, cout_t, 0, dev(xin), cin_t, dw, n, 2 * hs, 2 * hs, cout_t, cout_t, cin_t, 3, 2, 0, ws)
    assert rel_l2(host(dw), ref.numpy()) < TOL


def test_conv2d_transpose_dgrad_via_fwd():
    """dgrad of Conv2DTranspose = stride-2 SAME conv of dz with the transposed Keras kernel."""
    ops = _ops()
    rng = np.random.default_rng(8)
    n, h, cin, cout = 2, 4, 32, 64
    w = rng.standard_normal((3, 3, cout, cin)) * 0.1
    dz = rng.standard_normal((n, 2 * h, 2 * h, cout))
    xt = torch.zeros(n, cin, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_transpose_same(xt, t64(w)), xt, nchw(dz))
    wk = torch.zeros(9 * cin * cout, device="cuda")
    ops.transpose_taps(dev(w), wk, 9, cout, cin, cout)
    dx = torch.empty((n, h, h, cin), device="cuda")
    ops.conv2d_fwd(dev(dz), None, 0, cout, 0, wk, None, dx, cin, n, 2 * h, 2 * h, cout, cin, 3, 2, 1.0)
    assert rel_l2(host(dx), nhwc(ref)) < TOL


@pytest.mark.parametrize("n,h,c", [(2, 16, 64), (3, 8, 16), (1, 32, 128), (5, 2, 1024), (2, 4, 48)])
def test_instance_norm_fwd_bwd(n, h, c):
    ops = _ops()
    rng = np.random.default_rng(9)
    z = rng.standard_normal((n, h, h, c)) * 2 + 0.5
    a = np.where(z > 0, z, 0.2 * z)
    beta = rng.standard_normal(c) * 0.02
    g1 = rng.standard_normal((n, h, h, c))
    g2 = rng.standard_normal((n, h // 2, h // 2, c))
    zt = nchw(z).requires_grad_(True)
    at = F.leaky_relu(zt, 0.2)
    yt = st.instance_norm(at, t64(beta))
    pooled = F.avg_pool2d(yt, 2)
    ref_dz, = torch.autograd.grad([yt, pooled], [zt], [nchw(g1), nchw(g2)])
    stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    ad = dev(a)
    out = torch.empty((n, h, h, c), device="cuda")
    ops.in_stats(ad, c, stats, n, h * h, c, 1e-6)
    ops.in_apply(ad, c, stats, dev(beta), out, c, n, h * h, c)
    assert rel_l2(host(out), nhwc(yt.detach())) < TOL
    red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    dz = torch.empty((n, h, h, c), device="cuda")
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    ops.in_bwd(dev(g1), c, dev(g2), c, ad, c, stats, red, dz, c, db, n, h, h, c, 0.2)
    assert rel_l2(host(dz), nhwc(ref_dz)) < TOL
    assert rel_l2(host(db), nhwc(ref_dz).sum(axis=(0, 1, 2))) < TOL
    # pooling forward
    pl = torch.empty((n, h // 2, h // 2, c), device="cuda")
    ops.avgpool2_fwd(out, c, pl, c, n, h, h, c)
    assert rel_l2(host(pl), nhwc(pooled.detach())) < TOL
    # the same two results from ONE pass (shm_in_apply_pool): bit-identical, in both dtypes
    for dt in (torch.float32, torch.bfloat16):
        a_t = ad.to(dt)
        o1, p1 = torch.empty((n, h, h, c), device="cuda", dtype=dt), torch.empty((n, h // 2, h // 2, c), device="cuda", dtype=dt)
        o2, p2 = torch.full_like(o1, 7.0), torch.full_like(p1, 7.0)
        ops.in_apply(a_t, c, stats, dev(beta), o1, c, n, h * h, c)
        ops.avgpool2_fwd(o1, c, p1, c, n, h, h, c)
        ops.in_apply_pool(a_t, c, stats, dev(beta), o2, c, p2, c, n, h, h, c)
        assert torch.equal(o1, o2) and torch.equal(p1, p2)
    # numpy restatement of the backward (no pooled term)
    ops.in_bwd(dev(g1), c, None, 0, ad, c, stats, red, dz, c, None, n, h, h, c, 0.2)
    ref2 = tn.leaky_relu_grad(z, tn.instance_norm_bwd(a, g1))
    assert rel_l2(host(dz), ref2) < TOL


@pytest.mark.parametrize("n,h,c", [(3, 16, 64), (2, 12, 16), (1, 32, 128)])
def test_head_on_the_unnormalised_activation(n, h, c):
    """shm_head_in_fwd / shm_head_in_bwd (InstanceNorm apply of the block in front of the head folded into the head) against the
    two-pass form shm_in_apply + shm_head_fwd / shm_head_bwd: fp32 forward and input gradient bit-identical, the weight / bias
    gradients to 1e-6; bf16 (no rounding of the normalised value in between) within one bf16 rounding."""
    ops = _ops()
    rng = np.random.default_rng(17)
    z = rng.standard_normal((n, h, h, c)) * 2 + 0.5
    a = np.where(z > 0, z, 0.2 * z)
    beta = rng.standard_normal(c) * 0.02
    w = rng.standard_normal(c) * 0.1
    b = np.array([0.3])
    g = rng.standard_normal((n, h, h, 1))
    for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 8e-3)):
        ad = dev(a).to(dt)
        stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
        ops.in_stats(ad, c, stats, n, h * h, c, 1e-6)
        ahat = torch.empty((n, h, h, c), device="cuda", dtype=dt)
        ops.in_apply(ad, c, stats, dev(beta), ahat, c, n, h * h, c)
        res = []
        for fused in (False, True):
            y = torch.empty((n, h, h, 1), device="cuda")
            dx = torch.empty((n, h, h, c), device="cuda", dtype=dt)
            dwa = torch.zeros(c, dtype=torch.float64, device="cuda")
            dba = torch.zeros(1, dtype=torch.float64, device="cuda")
            red = torch.zeros(ops.LRELU_RED_SLOTS * (c + 1), dtype=torch.float64, device="cuda")
            if fused:
                ops.head_in_fwd(ad, c, stats, dev(beta), dev(w), dev(b), y, n, h * h, c, 0.2)
                ops.head_in_bwd(ad, c, stats, dev(beta), dev(w), y, dev(g), dx, c, dwa, dba, n, h * h, c, 0.2, red)
            else:
                ops.head_fwd(ahat, c, dev(w), dev(b), y, n * h * h, c, 0.2)
                ops.head_bwd(ahat, c, dev(w), y, dev(g), dx, c, dwa, dba, n * h * h, c, 0.2, red)
            res.append((host(y), host(dx.float()), host(dwa), host(dba)))
        # the backward of the block in front of the head from the rank-1 factors (shm_in_bwd_rank1) against shm_in_bwd on the
        # materialised gradient: fp32 to 1e-6, bf16 within the rounding of the gradient tensor it no longer stores
        dxm = torch.empty((n, h, h, c), device="cuda", dtype=dt)
        hdz = torch.empty((n, h, h), device="cuda")
        dwa = torch.zeros(c, dtype=torch.float64, device="cuda")
        dba = torch.zeros(1, dtype=torch.float64, device="cuda")
        red = torch.zeros(ops.LRELU_RED_SLOTS * (c + 1), dtype=torch.float64, device="cuda")
        yh = torch.empty((n, h, h, 1), device="cuda")
        ops.head_in_fwd(ad, c, stats, dev(beta), dev(w), dev(b), yh, n, h * h, c, 0.2)
        ops.head_in_bwd(ad, c, stats, dev(beta), dev(w), yh, dev(g), dxm, c, dwa, dba, n, h * h, c, 0.2, red, dz_out=hdz)
        dwb, dbb = torch.zeros_like(dwa), torch.zeros_like(dba)
        ops.head_in_bwd(ad, c, stats, dev(beta), dev(w), yh, dev(g), None, 0, dwb, dbb, n, h * h, c, 0.2, red, dz_out=hdz)
        assert rel_l2(host(dwb), host(dwa)) < 1e-12 and rel_l2(host(dbb), host(dba)) < 1e-12
        r3 = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
        dz_a, dz_b = torch.empty((n, h, h, c), device="cuda", dtype=dt), torch.empty((n, h, h, c), device="cuda", dtype=dt)
        db_a, db_b = torch.zeros(c, dtype=torch.float64, device="cuda"), torch.zeros(c, dtype=torch.float64, device="cuda")
        ops.in_bwd(dxm, c, None, 0, ad, c, stats, r3, dz_a, c, db_a, n, h, h, c, 0.2)
        ops.in_bwd_rank1(hdz, dev(w), ad, c, stats, r3, dz_b, c, db_b, n, h, h, c, 0.2)
        assert float(r3.abs().max()) == 0.0
        if tol == 0.0:
            # (not bitwise: with the product formed in registers hipcc contracts it into the sums' FMAs)
            assert rel_l2(host(dz_b), host(dz_a)) < 1e-6 and rel_l2(host(db_b), host(db_a)) < 1e-6
        else:
            assert rel_l2(host(dz_b.float()), host(dz_a.float())) < 2e-2
        (y0, dx0, w0, b0), (y1, dx1, w1, b1) = res
        if tol == 0.0:
            assert np.array_equal(y0, y1) and np.array_equal(dx0, dx1)
            assert rel_l2(w1, w0) < 1e-6 and rel_l2(b1, b0) < 1e-6          # four-pixel fp32 partial sums, grouped per sample here
        else:           # bf16: the two forms round differently, so a logit next to zero may take the other LeakyReLU slope
            same = ((y0 > 0) == (y1 > 0))[..., 0]
            # (the weight / bias sums then move by whole pixels: the bf16 head gradients are held to the oracle, with the masks
            # pinned, by the whole-step tests of test_bf16_gpu.py)
            assert same.mean() > 0.98 and rel_l2(y1, y0) < tol and rel_l2(dx1[same], dx0[same]) < tol


def test_lrelu_bwd_head_patch_dense_mask():
    ops = _ops()
    rng = np.random.default_rng(10)
    n, h, c = 3, 8, 64
    # lrelu_bwd
    y = rng.standard_normal((n, h, h, c))
    dy = rng.standard_normal((n, h, h, c))
    dz = torch.empty((n, h, h, c), device="cuda")
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    ops.lrelu_bwd(dev(dy), c, dev(y), c, dz, c, db, n * h * h, c, 0.2, torch.zeros(64 * c, dtype=torch.float64, device="cuda"))
    ref = np.where(y > 0, dy, 0.2 * dy)
    assert rel_l2(host(dz), ref) < 1e-6 and rel_l2(host(db), ref.sum(axis=(0, 1, 2))) < TOL
    # head 1x1 conv -> 1 channel
    x = rng.standard_normal((n, h, h, c))
    w = rng.standard_normal(c) * 0.1
    b = np.array([0.3])
    xt, wt, bt = t64(x).requires_grad_(True), t64(w).requires_grad_(True), t64(b).requires_grad_(True)
    yt = F.leaky_relu((xt * wt).sum(-1, keepdim=True) + bt, 0.2)
    g = rng.standard_normal((n, h, h, 1))
    rdx, rdw, rdb = torch.autograd.grad(yt, [xt, wt, bt], t64(g))
    yd = torch.empty((n, h, h, 1), device="cuda")
    ops.head_fwd(dev(x), c, dev(w), dev(b), yd, n * h * h, c, 0.2)
    assert rel_l2(host(yd), yt.detach().numpy()) < TOL
    dx = torch.empty((n, h, h, c), device="cuda")
    dwa = torch.zeros(c, dtype=torch.float64, device="cuda")
    dba = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.head_bwd(dev(x), c, dev(w), yd, dev(g), dx, c, dwa, dba, n * h * h, c, 0.2)
    assert rel_l2(host(dx), rdx.numpy()) < TOL and rel_l2(host(dwa), rdw.numpy()) < TOL
    assert rel_l2(host(dba), rdb.numpy()) < TOL
    # patch 3x3 conv -> 1 channel
    hp, cp = 4, 256
    x = rng.standard_normal((n, hp, hp, cp))
    w = rng.standard_normal((3, 3, cp, 1)) * 0.05
    xt = nchw(x).requires_grad_(True)
    wo = t64(w).permute(3, 2, 0, 1).contiguous().requires_grad_(True)      # OIHW leaf (Cout = 1)
    yt = F.leaky_relu(F.conv2d(xt, wo, padding=1), 0.2)
    g = rng.standard_normal((n, hp, hp, 1))
    rdx, rdw = torch.autograd.grad(yt, [xt, wo], nchw(g))
    rdw = rdw.permute(2, 3, 1, 0)
    yd = torch.empty((n, hp, hp, 1), device="cuda")
    ops.patch_fwd(dev(x), cp, dev(w), yd, n, hp, hp, cp, 0.2)
    assert rel_l2(host(yd), nhwc(yt.detach())) < TOL
    dzp = torch.empty((n, hp, hp, 1), device="cuda")
    dx = torch.empty((n, hp, hp, cp), device="cuda")
    dw = torch.empty((3, 3, cp, 1), device="cuda")
    ops.patch_bwd(dev(x), cp, dev(w), yd, dev(g), dzp, dx, cp, dw, n, hp, hp, cp, 0.2)
    assert rel_l2(host(dx), nhwc(rdx)) < TOL and rel_l2(host(dw), rdw.numpy()) < TOL
    # dense
    kdim = hp * hp * cp
    wd = rng.standard_normal((kdim, 5)) * 0.02
    xt, wt = t64(x.reshape(n, -1)).requires_grad_(True), t64(wd).requires_grad_(True)
    yt = xt @ wt
    g = rng.standard_normal((n, 5))
    rdx, rdw = torch.autograd.grad(yt, [xt, wt], t64(g))
    yd = torch.empty((n, 5), device="cuda")
    ops.dense_fwd(dev(x), dev(wd), yd, n, kdim, 5)
    assert rel_l2(host(yd), yt.detach().numpy()) < TOL
    dx = torch.zeros((n, kdim), device="cuda")
    dw = torch.empty((kdim, 5), device="cuda")
    ops.dense_bwd(dev(x), dev(wd), dev(g), dx, dw, n, kdim, 5)
    assert rel_l2(host(dx), rdx.numpy()) < TOL and rel_l2(host(dw), rdw.numpy()) < TOL
    # mask
    m = (rng.random(x.shape) > 0.2).astype(np.float32)
    out = torch.empty(x.shape, device="cuda")
    ops.mul_mask(dev(x), dev(m), out, x.size, 1.25)
    assert rel_l2(host(out), x.astype(np.float32) * m * 1.25) < 1e-6


def test_colour_and_inputs():
    ops = _ops()
    rng = np.random.default_rng(11)
    B, S = 2, 16
    npix = S * S
    orig = [rng.random((B, S, S, 3)) for _ in range(5)]
    ds, ds_d = [], []
    for o in orig:
        ref, sc = st.per_image_standardization(st.rgb_to_yuv(t64(o)))
        yuv = torch.empty((B, S, S, 3), device="cuda")
        acc = torch.empty(B * 2, dtype=torch.float64, device="cuda")
        scale = torch.empty(B, device="cuda")
        ops.rgb2yuv_std(dev(o), yuv, acc, scale, B, npix)
        assert rel_l2(host(yuv), ref.numpy()) < TOL and rel_l2(host(scale), sc.numpy()) < TOL
        ds.append(ref.numpy())
        ds_d.append(yuv)
    cb = torch.empty((B, S, S, 2), device="cuda")
    ops.avg_cbcr(ds_d, cb, B * npix)
    cref = sum(d[..., 1:] for d in ds) / 5.0
    assert rel_l2(host(cb), cref) < TOL
    flags = (True, False, False, True, False)
    fmask = sum(1 << k for k in range(5) if flags[k])
    gi = torch.empty((B, S, S, 16), device="cuda")
    ops.build_gen_input(ds_d, None, fmask, 0, gi, B, npix)
    ref = np.zeros((B, S, S, 16))
    for k in range(5):
        if not flags[k]:
            ref[..., k] = ds[k][..., 0]
    ref[..., 9] = 1
    assert rel_l2(host(gi), ref) < TOL
    geny = rng.standard_normal((B, S, S, 1))
    ci = torch.empty((5 * B, S, S, 16), device="cuda")
    ops.build_gen_input(ds_d, dev(geny), fmask, 1, ci, B, npix)
    ref = np.zeros((5, B, S, S, 16))
    for k in range(5):
        for j in range(5):
            if j != k:
                ref[k, ..., j] = geny[..., 0] if flags[j] else ds[j][..., 0]
        ref[k, ..., 5 + k] = 1
    assert rel_l2(host(ci), ref.reshape(5 * B, S, S, 16)) < TOL
    # cyclic input backward
    dcyc = rng.standard_normal((5 * B, S, S, 16))
    dgy = torch.zeros((B, S, S, 1), device="cuda")
    ops.cyc_input_bwd(dev(dcyc), fmask, dgy, B, npix)
    d5 = dcyc.reshape(5, B, S, S, 16)
    ref = sum(d5[k, ..., j] for k in range(5) for j in range(5) if j != k and flags[j])
    assert rel_l2(host(dgy)[..., 0], ref) < TOL
    # yuv -> rgb (+ noise into the padded D input)
    ycyc = rng.standard_normal((5 * B, S, S, 1))
    noise = rng.standard_normal((5 * B, S, S, 3)) * 0.1
    rgb = torch.empty((5 * B, S, S, 3), device="cuda")
    dp = torch.empty((5 * B, S, S, 16), device="cuda")
    ops.yuv2rgb(dev(ycyc), cb, dev(noise), rgb, dp, 5 * B, B, npix)
    yuv_full = np.concatenate([ycyc, np.tile(cref, (5, 1, 1, 1))], -1)
    ref = st.yuv_to_rgb(t64(yuv_full)).numpy()
    assert rel_l2(host(rgb), ref) < TOL
    assert rel_l2(host(dp)[..., :3], ref + noise) < TOL and float(host(dp)[..., 3:].max()) == 0.0
    dy = torch.ones((5 * B, S, S, 1), device="cuda")
    ops.rgb16_to_dy(dp, dy, 5 * B * npix, 1)
    assert rel_l2(host(dy)[..., 0], 1 + (ref + noise).sum(-1)) < TOL


def test_adam_clip():
    ops = _ops()
    rng = np.random.default_rng(12)
    n = 10007
    w, m, v = rng.standard_normal(n), rng.standard_normal(n) * 0.1, rng.random(n) * 0.01
    g = rng.standard_normal(n) * 2
    rw, rm, rv = tn.adam_update(w, m, v, g * 0.5, 3, 2e-5, 0.5, 0.99)
    alpha = tn.exp_decay_lr(2e-5, 3) * np.sqrt(1 - 0.99 ** 4) / (1 - 0.5 ** 4)
    wd, md, vd = dev(w), dev(m), dev(v)
    ops.adam_clip(wd, md, vd, dev(g), n, float(alpha), 0.5, 0.99, 1e-7, 0.5)
    assert rel_l2(host(wd), rw) < 1e-6 and rel_l2(host(md), rm) < 1e-6 and rel_l2(host(vd), rv) < 1e-6


def test_error_reporting():
    from shmgan_amd._lib import ShmError
    ops = _ops()
    x = torch.zeros((1, 4, 4, 24), device="cuda")
    with pytest.raises(ShmError):
        ops.conv2d_fwd(x, None, 0, 24, 0, x, None, x, 24, 1, 4, 4, 24, 16, 3, 1, 1.0)   # cin % 16 != 0
    with pytest.raises(ShmError):
        ops.conv2d_fwd(x, None, 0, 32, 0, x, None, x, 32, 1, 4, 4, 32, 16, 5, 1, 1.0)   # ksize 5


# ------------------------------------------------------------------ SpecSeg building blocks
@pytest.mark.parametrize("n,h,cin,cout", [(2, 8, 32, 16), (1, 16, 256, 128), (3, 4, 64, 32)])
def test_conv2d_transpose2x2_fwd(n, h, cin, cout):
    ops = _ops()
    rng = np.random.default_rng(21)
    x = rng.standard_normal((n, h, h, cin))
    w = rng.standard_normal((2, 2, cout, cin)) * 0.1
    b = rng.standard_normal(cout)
    ref = F.conv_transpose2d(nchw(x), t64(w).permute(3, 2, 0, 1).contiguous(), stride=2)
    ref = nhwc(ref) + b
    y = torch.empty((n, 2 * h, 2 * h, cout), device="cuda")
    ops.conv2d_transpose2x2_fwd(dev(x), cin, dev(w), dev(b), y, cout, n, h, h, cin, cout, 1.0)
    assert rel_l2(host(y), ref) < TOL


@pytest.mark.parametrize("n,h,cin,cout", [(2, 16, 16, 16), (1, 8, 48, 32)])
def test_conv2d_relu_narrow(n, h, cin, cout):
    """slope 0 == ReLU on 16/32-wide layers (SpecSeg.py:34-36)."""
    ops = _ops()
    rng = np.random.default_rng(22)
    x = rng.standard_normal((n, h, h, cin))
    w = rng.standard_normal((3, 3, cin, cout)) * 0.1
    b = rng.standard_normal(cout)
    ref = np.maximum(conv_ref(x, w, 1) + b, 0.0)
    y = torch.empty((n, h, h, cout), device="cuda")
    ops.conv2d_fwd(dev(x), None, 0, cin, 0, _wk(w, cin), dev(b), y, cout, n, h, h, cin, cout, 3, 1, 0.0)
    assert rel_l2(host(y), ref) < TOL
    assert (host(y) >= 0).all()


def test_bn_apply_maxpool_pack_sigmoid():
    ops = _ops()
    rng = np.random.default_rng(23)
    n, h, c = 2, 8, 32
    a = rng.standard_normal((n, h, h, c))
    g, be, mu = rng.standard_normal(c), rng.standard_normal(c), rng.standard_normal(c)
    var = rng.random(c) + 0.1
    out = torch.empty((n, h, h, c), device="cuda")
    ops.bn_apply(dev(a), c, dev(g), dev(be), dev(mu), dev(var), 1e-3, out, c, n * h * h, c)
    ref = (a - mu) * g / np.sqrt(var + 1e-3) + be
    assert rel_l2(host(out), ref) < TOL
    p = torch.empty((n, h // 2, h // 2, c), device="cuda")
    ops.maxpool2_fwd(out, c, p, c, n, h, h, c)
    refp = host(out).reshape(n, h // 2, 2, h // 2, 2, c).max(axis=(2, 4))
    assert np.array_equal(host(p), refp)
    src = rng.standard_normal((n, h, h, 3))
    d16 = torch.full((n, h, h, 16), 7.0, device="cuda")
    ops.pack_channels(dev(src), 3, 1, 2, d16, 16, n * h * h)
    assert np.array_equal(host(d16)[..., :2], dev(src).cpu().numpy()[..., 1:3]) and (host(d16)[..., 2:] == 0).all()
    w, b = rng.standard_normal(c) * 0.3, rng.standard_normal(1)
    y = torch.empty((n, h, h, 1), device="cuda")
    ops.head_sigmoid_fwd(out, c, dev(w), dev(b), y, n * h * h, c)
    refy = 1.0 / (1.0 + np.exp(-(host(out) @ w + b)))
    assert np.abs(host(y)[..., 0] - refy).max() < 1e-6


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("nk,batch,h,cin,c,stride,mask", [(5, 2, 16, 10, 64, 1, 0b10110), (1, 3, 16, 3, 16, 2, 0b111), (2, 1, 8, 10, 32, 1, 0b00001)])
def test_first_layer_dgrad_channel_sum(dt, nk, batch, h, cin, c, stride, mask):
    """shm_sum_input_channels + shm_conv3x3_dgrad_sum1 == channel sum of the ordinary input gradient."""
    ops = _ops()
    rng = np.random.default_rng(31)
    w = rng.standard_normal((3, 3, cin, c)) * 0.1
    ho = -(-h // stride)
    dz = rng.standard_normal((nk * batch, ho, ho, c))
    if dt == "bf16":
        dz = torch.from_numpy(dz.astype(np.float32)).to(torch.bfloat16).double().numpy()
    masks = [(mask << k | mask >> (5 - k)) & 0b11111 if nk > 1 else mask for k in range(nk)]
    xt = torch.zeros(nk * batch, cin, h, h, dtype=torch.float64, requires_grad=True)
    full, = torch.autograd.grad(st.conv2d_same(xt, t64(w), stride), xt, nchw(dz))
    full = nhwc(full).reshape(nk, batch, h, h, cin)
    ref = np.zeros((batch, h, h))
    for k in range(nk):
        for j in range(cin):
            if (masks[k] >> j) & 1:
                ref += full[k, ..., j]
    weff = torch.empty((nk, 9, c), device="cuda")
    wd = dev(w)
    for k in range(nk):
        ops.sum_input_channels(wd, cin, c, masks[k], weff[k])
    out = torch.full((batch, h, h, 1), 0.5, device="cuda")
    dzd = dev(dz).to(torch.bfloat16) if dt == "bf16" else dev(dz)
    ops.conv3x3_dgrad_sum1(dzd, c, weff, out, nk, batch, h, h, c, stride, 1)
    assert rel_l2(host(out)[..., 0] - 0.5, ref) < 1e-5
    ops.conv3x3_dgrad_sum1(dzd, c, weff, out, nk, batch, h, h, c, stride, 0)
    assert rel_l2(host(out)[..., 0], ref) < 1e-5


def test_device_random_draws():
    """shm_randn / shm_keep_mask (GaussianNoise(0.1) SHM.py:352, Dropout(0.2) SHM.py:363) against the NumPy restatement of the
    same Philox-4x32-10 streams (oracle/rng_np.py, pinned by Random123's known answers): keep masks bit for bit, normals to
    float rounding of log / sincos; plus moments, reproducibility and stream independence at 4 M draws."""
    from oracle import rng_np
    ops = _ops()
    for n, seed, stream in ((4096, 1234, 0), (999, (25 << 32) | 3, 5)):
        t = torch.full((n + 2,), 7.0, device="cuda")
        ops.randn(t[:n], 0.1, seed, stream)
        assert np.abs(host(t[:n]) - rng_np.randn(n, 0.1, seed, stream)).max() < 2e-6
        assert float(t[n]) == 7.0 and float(t[n + 1]) == 7.0             # a length that is not a multiple of four stays in bounds
        ops.keep_mask(t[:n], 0.2, seed, stream)
        assert np.array_equal(host(t[:n]), rng_np.keep_mask(n, 0.2, seed, stream))
    n = 1 << 22
    a = torch.empty(n, device="cuda")
    ops.randn(a, 0.1, 1234, 0)
    x = host(a)
    assert abs(x.mean()) < 3e-4 and abs(x.std() - 0.1) < 3e-4
    assert abs(((x / 0.1) ** 4).mean() - 3.0) < 0.05 and np.abs(x).max() < 0.1 * 6.5          # normal kurtosis, bounded tails
    b = torch.empty(n, device="cuda")
    ops.randn(b, 0.1, 1234, 0)
    assert torch.equal(a, b)
    ops.randn(b, 0.1, 1234, 2)
    assert abs(np.corrcoef(x, host(b))[0, 1]) < 3e-3
    ops.randn(b, 0.1, 1235, 0)
    assert abs(np.corrcoef(x, host(b))[0, 1]) < 3e-3
    m = torch.empty(n, device="cuda")
    ops.keep_mask(m, 0.2, 99, 1)
    k = host(m)
    assert set(np.unique(k)) == {0.0, 1.0} and abs(k.mean() - 0.8) < 1e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_transpose_taps_batch_equals_the_per_layer_calls(dt):
    """shm_transpose_taps_multi (every layer of a model in one launch) against shm_transpose_taps per layer: identical bits,
    including the zero padding of the row pitch and shapes that are not multiples of the 32 x 32 tile."""
    ops = _ops()
    rng = np.random.default_rng(5)
    shapes = [(9, 10, 64, 16 if dt == torch.float32 else 32), (9, 64, 64, 64), (1, 512, 512, 512), (9, 33, 70, 48), (4, 128, 256, 128), (9, 1, 64, 32)]
    items, ref = [], []
    for ntaps, rows, cols, rp in shapes:
        w = torch.from_numpy(rng.standard_normal((ntaps, rows, cols)).astype(np.float32)).cuda()
        a = torch.full((ntaps * cols * rp,), 3.0, device="cuda", dtype=dt)
        b = torch.full((ntaps * cols * rp,), 5.0, device="cuda", dtype=dt)
        ops.transpose_taps(w, a, ntaps, rows, cols, rp)
        items.append((w, b, ntaps, rows, cols, rp))
        ref.append(a)
    ops.TransposeBatch(items).run()
    torch.cuda.synchronize()
    for (w, b, ntaps, rows, cols, rp), a in zip(items, ref):
        assert torch.equal(a, b), (ntaps, rows, cols, rp)
        got = b.float().reshape(ntaps, cols, rp)
        assert float(got[..., rows:].abs().max() if rp > rows else 0.0) == 0.0
