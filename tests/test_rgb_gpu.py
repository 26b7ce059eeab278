"""The discriminator's first layer on the compact image layout (csrc/conv_rgb.hip): Conv2D(3x3, stride 2, 'same') + LeakyReLU +
InstanceNorm statistics of a 3-channel image stored one 16-byte chunk per pixel, and its weight gradient
(/root/reference/ShmGANwithSSpecSeg.py:353, 386-389: Conv_LReLU_IN of the first discriminator block).

Checked against the float64 oracle convolution, and the generic tap-GEMM / split-K kernels are run on the same compact buffers
(a forced variant keeps them: they read K channels per tap across the neighbouring pixels, times the zero weight columns)."""
import numpy as np
import pytest
import torch

from oracle import step_torch as st
from util import conv_ref, dev, host, nchw, pad_c, rel_l2

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _ops():
    from shmgan_amd import ops
    return ops


def rb(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(BF).double().numpy()


def _wk(w_hwio, cin_pad, dt):
    ops = _ops()
    k, _, cin, cout = w_hwio.shape
    wt = torch.zeros(k * k * cout * cin_pad, device="cuda", dtype=dt)
    ops.transpose_taps(dev(w_hwio), wt, k * k, cin, cout, cin_pad)
    return wt


def _x(x, dt):
    """[n, h, w, 3] -> the compact layout: one 16-byte chunk per pixel"""
    return dev(pad_c(x, 4 if dt == torch.float32 else 8)).to(dt).contiguous()


# (12 x 256 x 256: a grid that fills the chip -- the size at which a store-data hazard of the first version showed, see tools/check_isa_hazards.py)
CASES = [(3, 32, 64, True), (2, 64, 32, False), (1, 32, 16, True), (5, 96, 48, False), (7, 32, 64, False), (12, 256, 64, False)]


@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("n,h,cout,with_bias", CASES)
def test_first_layer_forward_on_the_compact_image(dt, n, h, cout, with_bias):
    ops = _ops()
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, h, h, 3))
    w = rng.standard_normal((3, 3, 3, cout)) * 0.3
    b = rng.standard_normal(cout) if with_bias else None
    q = (lambda a: a) if dt == torch.float32 else rb
    ref = conv_ref(q(x), q(w), 2) + (b if with_bias else 0.0)
    ref = np.where(ref > 0, ref, 0.2 * ref)
    ho = h // 2
    kpad = 16 if dt == torch.float32 else 32
    wk, xd = _wk(w, kpad, dt), _x(x, dt)
    bias = dev(b) if with_bias else None
    tol = 1e-5 if dt == torch.float32 else 4e-3
    try:
        y = torch.full((n, ho, ho, cout), 7.0, device="cuda", dtype=dt)
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda") if n % 2 else None
        ops.conv2d_in_fwd(xd, None, 0, xd.shape[-1], 0, wk, bias, y, cout, n, h, h, kpad, cout, 3, 2, 0.2, stats, 1e-6, scratch=scr)
        assert ops.last_kernel().startswith("conv3x3s2_rgb_fwd_kernel<"), ops.last_kernel()
        assert rel_l2(host(y.float()), ref) < tol
        yy = host(y.float()).astype(np.float64).reshape(n, -1, cout)
        s = host(stats).reshape(n, cout, 2)
        assert np.abs(s[..., 0] - yy.mean(1)).max() < 1e-4
        assert np.abs(s[..., 1] - 1.0 / np.sqrt(yy.var(1) + 1e-6)).max() < 1e-3 * s[..., 1].max()
        # without statistics, slope 1 (no activation)
        y1 = torch.empty_like(y)
        ops.conv2d_fwd(xd, None, 0, xd.shape[-1], 0, wk, bias, y1, cout, n, h, h, kpad, cout, 3, 2, 1.0)
        assert ops.last_kernel().startswith("conv3x3s2_rgb_fwd_kernel<")
        assert rel_l2(host(y1.float()), conv_ref(q(x), q(w), 2) + (b if with_bias else 0.0)) < tol
        # into a wider tensor (output pitch > cout: the kernel's direct stores instead of the linear ones out of LDS)
        yw = torch.full((n, ho, ho, 2 * cout), 7.0, device="cuda", dtype=dt)
        ops.conv2d_in_fwd(xd, None, 0, xd.shape[-1], 0, wk, bias, yw, 2 * cout, n, h, h, kpad, cout, 3, 2, 0.2, stats, 1e-6, scratch=scr)
        assert ops.last_kernel().startswith("conv3x3s2_rgb_fwd_kernel<")
        assert (yw[..., :cout] == y).all() and (yw[..., cout:] == 7.0).all()
        # the generic kernels on the same compact buffer
        ops.set_tuning("tapgemm.variant", "dma128x64")
        y2 = torch.empty_like(y)
        stats2 = torch.empty_like(stats)
        ops.conv2d_in_fwd(xd, None, 0, xd.shape[-1], 0, wk, bias, y2, cout, n, h, h, kpad, cout, 3, 2, 0.2, stats2, 1e-6)
        assert ops.last_kernel().startswith("tapgemm_dma_kernel<"), ops.last_kernel()
        assert rel_l2(host(y2.float()), ref) < tol
        assert np.abs(host(stats2) - host(stats)).max() < 2e-3 * np.abs(host(stats)).max()
    finally:
        ops.set_tuning("reset", 0)


def test_first_layer_forward_propagates_nan_and_handles_odd_shapes_elsewhere():
    """a NaN pixel reaches exactly the outputs whose window holds it; a map the compact kernel does not take (width / 2 not a multiple of 16)
    runs on the generic kernels with the same buffers"""
    ops = _ops()
    rng = np.random.default_rng(12)
    n, h, cout = 2, 32, 64
    x = rng.standard_normal((n, h, h, 3))
    x[1, 10, 21, 1] = np.nan
    w = rng.standard_normal((3, 3, 3, cout)) * 0.3
    y = torch.empty((n, h // 2, h // 2, cout), device="cuda")
    xd = _x(x, torch.float32)
    ops.conv2d_fwd(xd, None, 0, 4, 0, _wk(w, 16, torch.float32), None, y, cout, n, h, h, 16, cout, 3, 2, 0.2)
    assert ops.last_kernel().startswith("conv3x3s2_rgb_fwd_kernel<")
    bad = np.isnan(host(y)).any(-1)
    want = np.zeros_like(bad)
    want[1, 4:6, 10:11] = True              # output (oh, ow) reads rows 2 oh .. 2 oh + 2, columns 2 ow .. 2 ow + 2
    assert (bad == want).all()
    h = 24                                  # 12 output columns
    x = rng.standard_normal((n, h, h, 3))
    y = torch.empty((n, h // 2, h // 2, cout), device="cuda")
    ops.conv2d_fwd(_x(x, torch.float32), None, 0, 4, 0, _wk(w, 16, torch.float32), None, y, cout, n, h, h, 16, cout, 3, 2, 0.2)
    assert not ops.last_kernel().startswith("conv3x3s2_rgb_fwd_kernel<")
    ref = conv_ref(x, w, 2)
    assert rel_l2(host(y), np.where(ref > 0, ref, 0.2 * ref)) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("n,h,cout", [(3, 64, 64), (2, 32, 16), (1, 32, 32), (5, 96, 48), (8, 128, 64), (2, 512, 64), (13, 256, 64)])
def test_first_layer_weight_gradient_on_the_compact_image(dt, n, h, cout):
    ops = _ops()
    rng = np.random.default_rng(13)
    x = rng.standard_normal((n, h, h, 3))
    ho = h // 2
    dy = rng.standard_normal((n, ho, ho, cout))
    q = (lambda a: a) if dt == torch.float32 else rb
    wt = torch.zeros(3, 3, 3, cout, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(nchw(q(x)), wt, 2), wt, nchw(q(dy)))
    kpad = 16 if dt == torch.float32 else 32
    xd, dz = _x(x, dt), dev(dy).to(dt)
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, ho, ho, 3, cout, 3) // 4 + 1024, device="cuda")
    try:
        dw = torch.full((3, 3, 3, cout), 3.0, device="cuda")
        ops.conv2d_wgrad(xd, None, 0, xd.shape[-1], 0, dz, cout, dw, n, h, h, 3, kpad, cout, 3, 2, 0, ws)
        assert ops.last_kernel().startswith("conv3x3s2_rgb_wgrad_kernel<"), ops.last_kernel()
        assert rel_l2(host(dw), ref.numpy()) < 1e-5           # float32 products and sums of (exactly representable) operands in both dtypes
        first = host(dw).copy()
        ops.conv2d_wgrad(xd, None, 0, xd.shape[-1], 0, dz, cout, dw, n, h, h, 3, kpad, cout, 3, 2, 1, ws)
        assert rel_l2(host(dw), 2 * ref.numpy()) < 1e-5
        dw2 = torch.empty_like(dw)
        ops.conv2d_wgrad(xd, None, 0, xd.shape[-1], 0, dz, cout, dw2, n, h, h, 3, kpad, cout, 3, 2, 0, ws)
        assert (host(dw2) == first).all()                      # fixed summation order: bitwise repeatable
        # the generic split-K kernel on the same compact buffer
        ops.set_tuning("wgrad.variant", 1)
        dw3 = torch.empty_like(dw)
        ops.conv2d_wgrad(xd, None, 0, xd.shape[-1], 0, dz, cout, dw3, n, h, h, 3, kpad, cout, 3, 2, 0, ws)
        assert not ops.last_kernel().startswith("conv3x3s2_rgb_wgrad_kernel<")
        assert rel_l2(host(dw3), ref.numpy()) < (1e-5 if dt == torch.float32 else 2e-5)
    finally:
        ops.set_tuning("reset", 0)


def test_first_layer_weight_gradient_of_other_widths_takes_the_generic_kernels():
    """an output row that is not a multiple of 16 pixels is not the compact kernels' shape: same buffers, generic kernels, same result"""
    ops = _ops()
    rng = np.random.default_rng(14)
    n, h, cout = 3, 40, 32
    x = rng.standard_normal((n, h, h, 3))
    dy = rng.standard_normal((n, h // 2, h // 2, cout))
    wt = torch.zeros(3, 3, 3, cout, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(nchw(x), wt, 2), wt, nchw(dy))
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, 3, cout, 3) // 4 + 1024, device="cuda")
    dw = torch.empty((3, 3, 3, cout), device="cuda")
    ops.conv2d_wgrad(_x(x, torch.float32), None, 0, 4, 0, dev(dy), cout, dw, n, h, h, 3, 16, cout, 3, 2, 0, ws)
    assert not ops.last_kernel().startswith("conv3x3s2_rgb_wgrad_kernel<")
    assert rel_l2(host(dw), ref.numpy()) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
def test_first_layer_at_the_step_size_by_properties(dt):
    """BASELINE configs[1]'s discriminator batch (96 images of 256 x 256; no oracle at this size in the GPU suite): the forward is linear in the
    image at slope 1 and agrees with the generic kernels on the same compact buffer; the fused statistics are the moments of the stored tensor;
    the weight gradient is linear in dz, agrees with the generic split-K kernel, and <dw, w> = <dz, conv(x, w)> (the adjoint identity)."""
    ops = _ops()
    torch.manual_seed(5)
    n, S, cout = 96, 256, 64
    pitch, kpad = (4, 16) if dt == torch.float32 else (8, 32)
    tol = 2e-6 if dt == torch.float32 else 1e-2
    xa = torch.zeros((n, S, S, pitch), device="cuda", dtype=dt)
    xb = torch.zeros_like(xa)
    xa[..., :3] = torch.randn((n, S, S, 3), device="cuda").to(dt)
    xb[..., :3] = torch.randn((n, S, S, 3), device="cuda").to(dt)
    w = torch.randn((3, 3, 3, cout), device="cuda") * 0.3
    wk = torch.zeros(9 * cout * kpad, device="cuda", dtype=dt)
    ops.transpose_taps(w, wk, 9, 3, cout, kpad)
    ho = S // 2

    def fwd(x, slope=1.0, stats=None):
        y = torch.empty((n, ho, ho, cout), device="cuda", dtype=dt)
        if stats is None:
            ops.conv2d_fwd(x, None, 0, pitch, 0, wk, None, y, cout, n, S, S, kpad, cout, 3, 2, slope)
        else:
            ops.conv2d_in_fwd(x, None, 0, pitch, 0, wk, None, y, cout, n, S, S, kpad, cout, 3, 2, slope, stats, 1e-6)
        return y
    try:
        ya, yb = fwd(xa), fwd(xb)
        assert ops.last_kernel().startswith("conv3x3s2_rgb_fwd_kernel<")
        ysum = fwd((xa.float() + xb.float()).to(dt))
        ref = ya.float() + yb.float()
        assert float((ysum.float() - ref).norm() / ref.norm()) < (tol if dt == torch.float32 else 2e-2)
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        yl = fwd(xa, 0.2, stats)
        st_ = stats.reshape(n, cout, 2)
        yy = yl.double().reshape(n, -1, cout)
        assert float((st_[..., 0] - yy.mean(1)).abs().max()) < 1e-4
        assert float((st_[..., 1] - 1.0 / torch.sqrt(yy.var(1, unbiased=False) + 1e-6)).abs().max()) < 1e-3 * float(st_[..., 1].max())
        ops.set_tuning("tapgemm.variant", "dma128x64")
        yg = fwd(xa, 0.2)
        assert not ops.last_kernel().startswith("conv3x3s2_rgb")
        ops.set_tuning("reset", 0)
        assert float((yg.float() - yl.float()).norm() / yl.float().norm()) < tol
        # weight gradient
        dza = torch.randn((n, ho, ho, cout), device="cuda").to(dt)
        dzb = torch.randn((n, ho, ho, cout), device="cuda").to(dt)
        ws = torch.empty(ops.conv2d_wgrad_workspace(n, ho, ho, 3, cout, 3) // 4 + 1024, device="cuda")

        def wg(dz):
            dw = torch.empty((3, 3, 3, cout), device="cuda")
            ops.conv2d_wgrad(xa, None, 0, pitch, 0, dz, cout, dw, n, S, S, 3, kpad, cout, 3, 2, 0, ws)
            return dw
        da, db = wg(dza), wg(dzb)
        assert ops.last_kernel().startswith("conv3x3s2_rgb_wgrad_kernel<")
        if dt == torch.float32:                 # (a bf16 sum of two dz tensors is rounded again: linearity only in fp32)
            dsum = wg(dza + dzb)
            assert float((dsum - (da + db)).norm() / (da + db).norm()) < 1e-5
        ops.set_tuning("wgrad.variant", 1)
        dg = wg(dza)
        assert not ops.last_kernel().startswith("conv3x3s2_rgb")
        ops.set_tuning("reset", 0)
        assert float((dg - da).norm() / da.norm()) < 1e-5
        # adjoint identity with the kernel's own forward (operands as stored: w through the K-padded copy's rounding in bf16)
        wq = w if dt == torch.float32 else w.to(BF).float()
        lhs = float((da.double() * wq.double()).sum())
        rhs = float((dza.double() * ya.double()).sum())
        assert abs(lhs - rhs) < (1e-5 if dt == torch.float32 else 5e-3) * abs(rhs)
    finally:
        ops.set_tuning("reset", 0)


def test_first_layer_degenerate_calls():
    """batch 0 is a no-op; an image of 2 GiB or more is not the compact kernels' (32-bit offsets inside an image) but still the generic ones'"""
    ops = _ops()
    w = torch.randn((3, 3, 3, 16), device="cuda")
    wk = torch.zeros(9 * 16 * 16, device="cuda")
    ops.transpose_taps(w, wk, 9, 3, 16, 16)
    x = torch.zeros((1, 32, 32, 4), device="cuda")
    y = torch.full((1, 16, 16, 16), 3.0, device="cuda")
    ops.conv2d_fwd(x, None, 0, 4, 0, wk, None, y, 16, 0, 32, 32, 16, 16, 3, 2, 0.2)
    torch.cuda.synchronize()
    assert float(y.min()) == 3.0
    dw = torch.full((3, 3, 3, 16), 5.0, device="cuda")
    ws = torch.empty(1 << 20, device="cuda")
    ops.conv2d_wgrad(x, None, 0, 4, 0, y, 16, dw, 1, 32, 32, 3, 16, 16, 3, 2, 0, ws)        # zero image: zero gradient, whatever dz
    assert ops.last_kernel().startswith("conv3x3s2_rgb_wgrad_kernel<") and float(dw.abs().max()) == 0.0


@pytest.mark.parametrize("dt", [torch.float32, BF], ids=["f32", "bf16"])
@pytest.mark.parametrize("n,hi,wi,cout", [(2, 64, 96, 64), (3, 96, 32, 32), (1, 32, 128, 16)])
def test_first_layer_on_non_square_images(dt, n, hi, wi, cout):
    """rows and columns are independent in the kernels' indexing: forward and weight gradient on rectangular images against the oracle"""
    ops = _ops()
    rng = np.random.default_rng(15)
    x = rng.standard_normal((n, hi, wi, 3))
    w = rng.standard_normal((3, 3, 3, cout)) * 0.3
    ho, wo = hi // 2, wi // 2
    dy = rng.standard_normal((n, ho, wo, cout))
    q = (lambda a: a) if dt == torch.float32 else rb
    ref = conv_ref(q(x), q(w), 2)
    ref = np.where(ref > 0, ref, 0.2 * ref)
    wt = torch.zeros(3, 3, 3, cout, dtype=torch.float64, requires_grad=True)
    gref, = torch.autograd.grad(st.conv2d_same(nchw(q(x)), wt, 2), wt, nchw(q(dy)))
    kpad = 16 if dt == torch.float32 else 32
    xd, wk = _x(x, dt), _wk(w, kpad, dt)
    y = torch.empty((n, ho, wo, cout), device="cuda", dtype=dt)
    stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
    ops.conv2d_in_fwd(xd, None, 0, xd.shape[-1], 0, wk, None, y, cout, n, hi, wi, kpad, cout, 3, 2, 0.2, stats, 1e-6)
    assert ops.last_kernel().startswith("conv3x3s2_rgb_fwd_kernel<"), ops.last_kernel()
    assert rel_l2(host(y.float()), ref) < (1e-5 if dt == torch.float32 else 4e-3)
    yy = host(y.float()).astype(np.float64).reshape(n, -1, cout)
    assert np.abs(host(stats).reshape(n, cout, 2)[..., 0] - yy.mean(1)).max() < 1e-4
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, ho, wo, 3, cout, 3) // 4 + 1024, device="cuda")
    dw = torch.empty((3, 3, 3, cout), device="cuda")
    ops.conv2d_wgrad(xd, None, 0, xd.shape[-1], 0, dev(dy).to(dt), cout, dw, n, hi, wi, 3, kpad, cout, 3, 2, 0, ws)
    assert ops.last_kernel().startswith("conv3x3s2_rgb_wgrad_kernel<"), ops.last_kernel()
    assert rel_l2(host(dw), gref.numpy()) < 1e-5
