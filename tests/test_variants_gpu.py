"""Every MFMA kernel variant the convolution entry points can dispatch to, against the float64 oracle.

The variant a call takes depends on shape, dtype and grid size, so the small shapes of test_ops_gpu.py only ever
reach some of them.  Here each variant is FORCED through the C ABI (shm_set_tuning("tapgemm.variant", SHM_TG_*));
a forced variant that the shape is not eligible for is an error, and shm_last_kernel() is checked, so a passing case
proves that variant computed the result.  The second half runs real layer shapes of BASELINE configs[1]
(S=256, F=64, B=8) through the DEFAULT dispatch and asserts which kernel took them.

Tolerances: fp32 rel-L2 <= 1e-5 (single conv op); bf16 operands rounded for the oracle, results
<= 4e-3 (one bf16 rounding), fused statistics as in test_bf16_gpu.py.
"""
import numpy as np
import pytest
import torch

from oracle import step_torch as st
from util import conv_ref, host, nchw, nhwc, rel_l2, t64

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
TOL = {"f32": 1e-5, "bf16": 4e-3}
SYMBOL = {
    "halo128": "tapgemm_halo_kernel<{t}, {t}, 128, 16, false, 2>", "halo64": "tapgemm_halo_kernel<{t}, {t}, 64, 16, false, 2>",
    "halo128_ph8": "tapgemm_halo_kernel<{t}, {t}, 128, 8, false, 2>",
    "dma128x128": "tapgemm_dma_kernel<{t}, {t}, 128, 128, 2, 2, 3, 16>", "dma64x128": "tapgemm_dma_kernel<{t}, {t}, 64, 128, 2, 2, 3, 16>",
    "dma128x64": "tapgemm_dma_kernel<{t}, {t}, 128, 64, 2, 2, 3, 16>", "dma64x64": "tapgemm_dma_kernel<{t}, {t}, 64, 64, 2, 2, 3, 16>", "dma256x64": "tapgemm_dma_kernel<{t}, {t}, 256, 64, 4, 1, 3, 16>",
    "dma256x128": "tapgemm_dma_kernel<{t}, {t}, 256, 128, 4, 2, 3, 16>",
    "dma128x128_bk32": "tapgemm_dma_kernel<{t}, {t}, 128, 128, 2, 2, 2, 32>",
    "dma128x128_nst4": "tapgemm_dma_kernel<{t}, {t}, 128, 128, 2, 2, 4, 16>",
    "wreg": "tapgemm_wreg16_bf16_kernel<{nch}, {epi}>",                            # bf16: the eight-wave kernel with line-wide stores ("tapgemm.wreg16", default on)
    "wreg4": "tapgemm_wreg_kernel<{t}, {nch}>",                             # ... and the four-wave form with 32-column wave tiles
    "halo128_st": "tapgemm_halo_kernel<{t}, {t}, 128, 16, true, 2>", "halo64_st": "tapgemm_halo_kernel<{t}, {t}, 64, 16, true, 2>",
    "phase4": "tapgemm_phase4_kernel<{t}, {t}>", "halo128_st_w4": "tapgemm_halo_kernel<{t}, {t}, 128, 16, true, 4>",
}
HALO = ["halo128", "halo64", "halo128_st", "halo64_st", "halo128_st_w4"]
DMA = ["dma128x128", "dma64x128", "dma128x64", "dma64x64", "dma256x64", "dma256x128", "dma128x128_bk32", "dma128x128_nst4"]


def _ops():
    from shmgan_amd import ops
    return ops


@pytest.fixture(autouse=True)
def _reset_tuning():
    yield
    _ops().set_tuning("reset", 0)


def _sym(variant, dt, nch=2, epi=True, hw=None):
    """epi: the launch has an activation or fused statistics (forward); False = the plain input-gradient form (round 5: tapgemm_wreg16_bf16_kernel<NCH, EPI>).
    hw = (height, width) of the map: under "tapgemm.wreg16" = 2 (the default) a 64-input-channel bf16 layer on a map of whole 8 x 32-pixel
    patches takes the ping-pong kernel (conv_pingpong.hip)."""
    if variant == "wreg" and dt == "f32":
        return f"tapgemm_wreg_f32_kernel<{int(2 * nch)}, 4, false>"          # 16-channel chunks, four N waves, one source
    nch = int(nch)
    if variant == "wreg" and SYMBOL["wreg"].startswith("tapgemm_wreg16") and nch == 2 and hw is not None and hw[0] % 8 == 0 and hw[1] % 32 == 0 \
            and _ops().get_tuning("tapgemm.wreg16") == 2:
        return f"tapgemm_pp_bf16_kernel<{2 if epi else 0}>"          # MODE 2: bias + LeakyReLU + statistics; 0: the plain product
    return SYMBOL[variant].format(t="float" if dt == "f32" else "__bf16", nch=nch, epi="true" if epi else "false")


def _dev(a, dt):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    return t.to(BF) if dt == "bf16" else t


def _rnd(a, dt):
    """the operand as the device holds it (float64)."""
    if dt == "bf16":
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(BF).double().numpy()
    return np.asarray(a, np.float32).astype(np.float64)


def _wk(w_hwio, cin_pad, dt):
    ops = _ops()
    k, _, cin, cout = w_hwio.shape
    wt = torch.zeros(k * k * cout * cin_pad, device="cuda", dtype=BF if dt == "bf16" else torch.float32)
    ops.transpose_taps(torch.from_numpy(np.ascontiguousarray(w_hwio, dtype=np.float32)).cuda(), wt, k * k, cin, cout, cin_pad)
    return wt


def _check_stats(stats, y, n, cout, dt):
    yy = y.reshape(n, -1, cout)
    s = host(stats).reshape(n, cout, 2)
    assert np.abs(s[..., 0] - yy.mean(1)).max() < (1e-4 if dt == "bf16" else 1e-6)
    ref_inv = 1.0 / np.sqrt(yy.var(1) + 1e-6)
    assert np.abs(s[..., 1] / ref_inv - 1).max() < (1e-3 if dt == "bf16" else 1e-5)


def _fwd_case(variant, dt, n, h, c1, c2, cout, k, s, seed=0):
    """Conv2D(+concat) + bias + LeakyReLU + fused InstanceNorm statistics under a forced variant."""
    ops = _ops()
    rng = np.random.default_rng(100 + seed)
    cin = c1 + c2
    xa = rng.standard_normal((n, h, h, c1))
    xb = rng.standard_normal((n, h, h, c2)) if c2 else None
    w = rng.standard_normal((k, k, cin, cout)) * 0.1
    b = rng.standard_normal(cout)
    xr = _rnd(xa, dt) if xb is None else np.concatenate([_rnd(xa, dt), _rnd(xb, dt)], -1)
    ref = conv_ref(xr, _rnd(w, dt), s) + b
    ref = np.where(ref > 0, ref, 0.2 * ref)
    ho = ref.shape[1]
    y = torch.full((n, ho, ho, cout), 9.0, device="cuda", dtype=BF if dt == "bf16" else torch.float32)
    stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
    ops.set_tuning("tapgemm.variant", variant)
    ops.conv2d_in_fwd(_dev(xa, dt), None if xb is None else _dev(xb, dt), c1 if c2 else 0, c1, c2, _wk(w, cin, dt),
                      torch.from_numpy(b.astype(np.float32)).cuda(), y, cout, n, h, h, cin, cout, k, s, 0.2, stats, 1e-6, scratch=scr)
    torch.cuda.synchronize()
    assert ops.last_kernel() == _sym(variant, dt, cin / 32, hw=(h, h)), ops.last_kernel()
    got = host(y.float())
    assert rel_l2(got, ref) < TOL[dt], (variant, dt, rel_l2(got, ref))
    if (ho * ho) % 64 == 0:                   # the fused path (smaller maps take a separate statistics pass)
        _check_stats(stats, got, n, cout, dt)
        assert float(scr.abs().max()) == 0.0  # "zero on entry, zero on return"


# ---- unit-stride 3x3 on maps that are multiples of 16: every variant is eligible
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant", HALO + DMA)
@pytest.mark.parametrize("n,h,c1,c2,cout", [
    (3, 16, 64, 0, 64),          # one tile wide
    (2, 32, 64, 0, 160),         # ragged N (160 = 128 + 32), several patches per image
    (2, 16, 64, 128, 128),       # Concatenate([up, skip]) read from two tensors
    (5, 16, 128, 0, 48),         # odd batch, Cout < 64 (ragged inside one tile)
])
def test_conv3x3_s1_forced_variant(variant, dt, n, h, c1, c2, cout):
    _fwd_case(variant, dt, n, h, c1, c2, cout, 3, 1)


# ---- outputs "larger than 4 GiB" ("tapgemm.flat_epilogue"): the 64-bit-address element stores of every kernel family, which real tests
# cannot reach by size (the default epilogues go through 32-bit buffer descriptors)
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant,n,h,c1,c2,cout,k,s", [
    ("halo128_st", 2, 32, 64, 0, 160, 3, 1), ("halo64_st", 2, 16, 64, 128, 128, 3, 1), ("dma128x128", 2, 32, 64, 0, 160, 3, 1),
    ("dma256x128", 3, 32, 64, 0, 128, 3, 2), ("dma64x128", 5, 16, 128, 0, 48, 3, 2), ("dma128x128_bk32", 2, 32, 64, 0, 128, 3, 2),
])
def test_flat_epilogue_forced_variant(variant, dt, n, h, c1, c2, cout, k, s):
    _ops().set_tuning("tapgemm.flat_epilogue", 1)
    _fwd_case(variant, dt, n, h, c1, c2, cout, k, s, seed=8)


def test_flat_epilogue_refuses_the_buffer_store_kernels_and_keeps_results():
    """the weights-in-registers kernels store through buffer descriptors: with outputs "beyond 4 GiB" the automatic choice moves on to
    another kernel and a forced one is refused; Conv2DTranspose (four-phase kernel) keeps its result through the 64-bit path"""
    ops = _ops()
    from shmgan_amd._lib import ShmError
    ops.set_tuning("tapgemm.flat_epilogue", 1)
    rng = np.random.default_rng(12)
    n, h, cin, cout = 2, 32, 64, 64
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    ref = conv_ref(x.astype(np.float64), w.astype(np.float64), 1)
    ref = np.where(ref > 0, ref, 0.2 * ref)
    y = torch.empty((n, h, h, cout), device="cuda")
    ops.conv2d_fwd(_dev(x, "f32"), None, 0, cin, 0, _wk(w, cin, "f32"), None, y, cout, n, h, h, cin, cout, 3, 1, 0.2)
    assert not ops.last_kernel().startswith("tapgemm_wreg"), ops.last_kernel()
    assert rel_l2(host(y), ref) < 1e-5
    ops.set_tuning("tapgemm.variant", "wreg")
    with pytest.raises(ShmError):
        ops.conv2d_fwd(_dev(x, "f32"), None, 0, cin, 0, _wk(w, cin, "f32"), None, y, cout, n, h, h, cin, cout, 3, 1, 0.2)
    ops.set_tuning("tapgemm.variant", "auto")
    ci, co, hs = 128, 64, 16
    xt = rng.standard_normal((n, hs, hs, ci)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, co, ci)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    r = nhwc(st.conv2d_transpose_same(nchw(xt.astype(np.float64)), t64(wt))) + b
    r = np.where(r > 0, r, 0.2 * r)
    yt = torch.empty((n, 2 * hs, 2 * hs, co), device="cuda")
    ops.set_tuning("tapgemm.phase4_min_blocks", 0)
    ops.conv2d_transpose_fwd(torch.from_numpy(xt).cuda(), ci, torch.from_numpy(wt).cuda(), torch.from_numpy(b).cuda(), yt, co, n, hs, hs, ci, co, 0.2)
    assert ops.last_kernel().startswith("tapgemm_phase4_kernel"), ops.last_kernel()
    assert rel_l2(host(yt), r) < 1e-5


# ---- weights-in-registers kernels (<= 64 input channels from one tensor): persistent blocks over 8 x 16 patches
@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("n,h,cin,cout", [
    (3, 16, 64, 64),        # 6 patches per image
    (2, 32, 64, 192),       # three N tiles
    (5, 16, 32, 128),       # the first-layer pitch: K = 32 in bf16 / 16 in fp32
    (1, 64, 64, 64),        # 32 patches in one image: several patches per block when the grid is capped
    (7, 48, 64, 64),        # 126 patches over 7 images: blocks that cross image boundaries (statistics flush)
])
def test_wreg_forced_variant(dt, n, h, cin, cout):
    if dt == "f32" and cin == 32:
        cin = 16
    _fwd_case("wreg", dt, n, h, cin, 0, cout, 3, 1, seed=3)
    if dt == "bf16":                          # "tapgemm.wreg16" = 1: tapgemm_wreg16_bf16_kernel on the shapes the ping-pong kernel takes by default
        _ops().set_tuning("tapgemm.wreg16", 1)
        _fwd_case("wreg", dt, n, h, cin, 0, cout, 3, 1, seed=3)


# ---- the ping-pong kernel (conv_pingpong.hip; "tapgemm.wreg16" = 2, default): 64 input channels, maps of whole 8 x 32-pixel patches
@pytest.mark.parametrize("n,hi,wi,cout,slope", [
    (3, 16, 32, 64, 0.2),         # two patches per image, one above the other: every halo column of the map is padding
    (5, 48, 64, 128, 0.2),        # 12 patches per image, two channel blocks, groups that cross image boundaries (statistics flush)
    (2, 32, 96, 64, 0.0),         # interior patches with neighbours on every side, ReLU
    (9, 16, 32, 192, 1.0),        # two patches per image, more groups than patches on a 256-CU grid, no activation (bias + statistics only)
    (40, 64, 64, 64, 0.2),        # 640 patches on 512 groups: uneven ranges
])
def test_pingpong_forward_with_statistics(n, hi, wi, cout, slope):
    ops = _ops()
    rng = np.random.default_rng(200 + n)
    cin = 64
    x = rng.standard_normal((n, hi, wi, cin))
    w = rng.standard_normal((3, 3, cin, cout)) * 0.1
    b = rng.standard_normal(cout)
    ref = conv_ref(_rnd(x, "bf16"), _rnd(w, "bf16"), 1) + b
    ref = np.where(ref > 0, ref, slope * ref)
    y = torch.full((n, hi, wi, cout), 9.0, device="cuda", dtype=BF)
    stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
    ops.set_tuning("tapgemm.variant", "wreg")
    for rep in range(2):                      # the second call finds the statistics scratch as the first one left it (zero on return)
        ops.conv2d_in_fwd(_dev(x, "bf16"), None, 0, cin, 0, _wk(w, cin, "bf16"), torch.from_numpy(b.astype(np.float32)).cuda(), y, cout, n, hi, wi, cin, cout, 3, 1,
                          slope, stats, 1e-6, scratch=scr)
        torch.cuda.synchronize()
        assert ops.last_kernel() == "tapgemm_pp_bf16_kernel<2>", ops.last_kernel()
        got = host(y.float())
        assert rel_l2(got, ref) < TOL["bf16"], rel_l2(got, ref)
        if (hi * wi) % 64 == 0:
            _check_stats(stats, got, n, cout, "bf16")
        y.fill_(9.0)
    # bias + activation without statistics: MODE 1
    ops.conv2d_fwd(_dev(x, "bf16"), None, 0, cin, 0, _wk(w, cin, "bf16"), torch.from_numpy(b.astype(np.float32)).cuda(), y, cout, n, hi, wi, cin, cout, 3, 1, slope)
    assert ops.last_kernel() == "tapgemm_pp_bf16_kernel<1>", ops.last_kernel()
    assert rel_l2(host(y.float()), ref) < TOL["bf16"]
    # the same product without bias, activation or statistics: MODE 0
    ops.conv2d_fwd(_dev(x, "bf16"), None, 0, cin, 0, _wk(w, cin, "bf16"), None, y, cout, n, hi, wi, cin, cout, 3, 1, 1.0)
    assert ops.last_kernel() == "tapgemm_pp_bf16_kernel<0>", ops.last_kernel()
    assert rel_l2(host(y.float()), conv_ref(_rnd(x, "bf16"), _rnd(w, "bf16"), 1)) < TOL["bf16"]


# ---- "tapgemm.wreg16" = 0: the four-wave bf16 form with 32-column wave tiles (the default is the eight-wave form, test_wreg_forced_variant)
@pytest.mark.parametrize("n,h,cin,cout", [(3, 16, 64, 64), (2, 32, 64, 192), (5, 16, 32, 128), (1, 64, 64, 64), (7, 48, 64, 64)])
def test_wreg_four_wave_bf16(n, h, cin, cout):
    ops = _ops()
    sym = SYMBOL["wreg"]
    SYMBOL["wreg"] = SYMBOL["wreg4"]
    try:
        ops.set_tuning("tapgemm.wreg16", 0)
        _fwd_case("wreg", "bf16", n, h, cin, 0, cout, 3, 1, seed=5)
    finally:
        SYMBOL["wreg"] = sym


def test_pingpong_repeat_launches_are_bitwise_identical():
    """The ping-pong kernel orders its LDS traffic by hand (inline-asm LDS accesses beside an LDS-DMA in flight, counted waits, four barriers per
    period): at the north star's size, which fills the chip, thirty launches of the forward block and of its input gradient must give the same
    bits -- outputs and fused statistics -- and agree with the eight-wave kernel ("tapgemm.wreg16" = 1) to bf16 rounding of a different fp32
    summation order.  (The race detector of round 2, tools/probes/conv_repeat_probe.py, as a test.)"""
    ops = _ops()
    n, h, c = 40, 256, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((n, h, h, c), device="cuda", generator=g).to(BF)
    w = torch.randn((3, 3, c, c), device="cuda", generator=g) * 0.05
    wk = torch.zeros(9 * c * c, device="cuda", dtype=BF)
    ops.transpose_taps(w, wk, 9, c, c, c)
    b = torch.randn(c, device="cuda", generator=g)
    y = torch.empty((n, h, h, c), device="cuda", dtype=BF)
    dx = torch.empty((n, h, h, c), device="cuda", dtype=BF)
    stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
    ops.set_tuning("tapgemm.variant", "wreg")
    first = None
    for rep in range(30):
        y.fill_(3.0)
        dx.fill_(3.0)
        ops.conv2d_in_fwd(x, None, 0, c, 0, wk, b, y, c, n, h, h, c, c, 3, 1, 0.2, stats, 1e-6, scratch=scr)
        assert ops.last_kernel() == "tapgemm_pp_bf16_kernel<2>"
        ops.conv2d_dgrad(y, c, w.to(BF), dx, None, c, c, 0, n, h, h, c, c, 3, 1)
        assert ops.last_kernel() == "tapgemm_pp_bf16_kernel<0>"
        torch.cuda.synchronize()
        cur = (y.clone(), stats.clone(), dx.clone())
        if first is None:
            first = cur
        else:
            # the statistics go through f64 atomics into a few slots: their order is not fixed, the sums agree to the last bits of a double
            assert torch.equal(cur[0], first[0]) and torch.equal(cur[2], first[2]), rep
            assert (cur[1] - first[1]).abs().max() <= 1e-12 * first[1].abs().max(), rep
    ops.set_tuning("tapgemm.wreg16", 1)
    ops.conv2d_in_fwd(x, None, 0, c, 0, wk, b, y, c, n, h, h, c, c, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    assert ops.last_kernel() == "tapgemm_wreg16_bf16_kernel<2, true>"
    torch.cuda.synchronize()
    d = (y.float() - first[0].float()).abs()
    assert float((d > 0).float().mean()) < 0.02 and float(d.max()) <= 0.0625 * float(first[0].float().abs().max())      # a bf16 ulp here and there
    assert (stats - first[1]).abs().max() <= 1e-3 * first[1].abs().max()


@pytest.mark.parametrize("dt,wreg16", [("f32", 1), ("bf16", 2), ("bf16", 1), ("bf16", 0)])
def test_wreg_kernels_keep_a_nan_a_nan(dt, wreg16):
    """Round-3 advisor finding: the two-instruction LeakyReLU of the weights-in-registers kernels (common.h: shm_lrelu_max) was
    v_med3(u, u * slope, FLT_MAX), which turns a NaN MFMA result into FLT_MAX -- a diverged activation became a finite number
    and the fused statistics stayed finite.  One NaN input pixel must poison its 3 x 3 output neighbourhood (every channel) and
    that sample's statistics, and nothing else."""
    ops = _ops()
    rng = np.random.default_rng(77)
    n, h, cin, cout = 2, 32, 64, 64
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    x[1, 10, 20, 5] = np.nan
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.1).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = torch.zeros((n, h, h, cout), device="cuda", dtype=BF if dt == "bf16" else torch.float32)
    stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
    ops.set_tuning("tapgemm.wreg16", wreg16)
    ops.set_tuning("tapgemm.variant", "wreg")
    ops.conv2d_in_fwd(_dev(x, dt), None, 0, cin, 0, _wk(w, cin, dt), torch.from_numpy(b).cuda(), y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    torch.cuda.synchronize()
    assert ops.last_kernel().startswith("tapgemm_pp" if wreg16 == 2 else "tapgemm_wreg"), ops.last_kernel()
    got = host(y.float())
    bad = np.isnan(got)
    want = np.zeros_like(bad)
    want[1, 9:12, 19:22, :] = True
    assert np.array_equal(bad, want), (int(bad.sum()), int(want.sum()))
    assert np.isfinite(got[~bad]).all() and np.abs(got[~bad]).max() < 1e3
    s = host(stats).reshape(n, cout, 2)
    assert np.isfinite(s[0]).all() and not np.isfinite(s[1][:, 0]).any()          # the poisoned sample's means


@pytest.mark.parametrize("n,h,c1,c2,cout,sym", [
    (2, 32, 16, 0, 16, "tapgemm_wreg_f32_kernel<1, 1, false>"),      # SpecSeg 256-level layers: 16 output channels, 32-row patches
    (3, 64, 16, 16, 16, "tapgemm_wreg_f32_kernel<2, 1, true>"),     # Concatenate([up, skip]) of two 16-channel tensors, several patches per image
    (1, 32, 32, 0, 16, "tapgemm_wreg_f32_kernel<2, 1, false>"),
    (2, 32, 16, 0, 32, "tapgemm_wreg_f32_kernel<1, 2, false>"),      # 32 output channels, 16-row patches
    (5, 16, 32, 0, 32, "tapgemm_wreg_f32_kernel<2, 2, false>"),
    (2, 32, 32, 0, 64, "tapgemm_wreg_f32_kernel<2, 4, false>"),         # K = 32, 64-channel blocks
])
def test_wreg_f32_narrow_and_concat(n, h, c1, c2, cout, sym):
    """The fp32 weights-in-registers kernel on 16 / 32 output channels (WN = 1 / 2 waves along N), K = 32, and SpecSeg's two-source form."""
    ops = _ops()
    rng = np.random.default_rng(31)
    cin = c1 + c2
    xa = rng.standard_normal((n, h, h, c1))
    xb = rng.standard_normal((n, h, h, c2)) if c2 else None
    w = rng.standard_normal((3, 3, cin, cout)) * 0.1
    b = rng.standard_normal(cout)
    xr = xa if xb is None else np.concatenate([xa, xb], -1)
    ref = np.maximum(conv_ref(xr.astype(np.float32), w.astype(np.float32), 1) + b, 0.0)       # ReLU (slope 0), as SpecSeg uses it
    y = torch.full((n, h, h, cout), 9.0, device="cuda")
    for variant in ("wreg", "auto"):
        ops.set_tuning("tapgemm.variant", variant)
        ops.conv2d_fwd(_dev(xa, "f32"), None if xb is None else _dev(xb, "f32"), c1 if c2 else 0, c1, c2, _wk(w, cin, "f32"),
                       torch.from_numpy(b.astype(np.float32)).cuda(), y, cout, n, h, h, cin, cout, 3, 1, 0.0)
        assert ops.last_kernel() == sym, ops.last_kernel()
        assert rel_l2(host(y), ref) < 1e-5
        y.fill_(9.0)


@pytest.mark.parametrize("dt", ["bf16", "f32"])
def test_wreg_many_patches_per_block_and_refusals(dt):
    """n = 24 at 128 x 128 is 3072 patches on 512 (bf16) / 256 (fp32) blocks: 6 / 12 patches per block, most blocks inside
    one image, some across two.  Channel counts the kernel has no instantiation for and concatenated inputs are refused."""
    from shmgan_amd._lib import ShmError
    ops = _ops()
    _fwd_case("wreg", dt, 24, 128, 64, 0, 64, 3, 1, seed=4)
    ops.set_tuning("tapgemm.variant", "wreg")
    adt = BF if dt == "bf16" else torch.float32
    kbad = 96 if dt == "bf16" else 48
    x = torch.zeros((1, 16, 16, kbad), device="cuda", dtype=adt)
    w = torch.zeros(9 * 64 * kbad, device="cuda", dtype=adt)
    y = torch.zeros((1, 16, 16, 64), device="cuda", dtype=adt)
    with pytest.raises(ShmError):
        ops.conv2d_fwd(x, None, 0, kbad, 0, w, None, y, 64, 1, 16, 16, kbad, 64, 3, 1, 1.0)
    x64 = torch.zeros((1, 16, 16, 64), device="cuda", dtype=adt)
    wb = torch.zeros(9 * 64 * 128, device="cuda", dtype=adt)
    with pytest.raises(ShmError):             # two sources
        ops.conv2d_fwd(x64, x64, 64, 64, 64, wb, None, y, 64, 1, 16, 16, 128, 64, 3, 1, 1.0)
    with pytest.raises(ShmError):             # Cout = 32
        ops.conv2d_fwd(x64, None, 0, 64, 0, wb, None, y, 32, 1, 16, 16, 64, 32, 3, 1, 1.0)


@pytest.mark.parametrize("variant", HALO)
def test_halo_ph8_and_halo_need_eligible_shape(variant):
    """halo variants refuse shapes they cannot take (no silent fallback); the 8-row patch variant is bf16 only."""
    from shmgan_amd._lib import ShmError
    ops = _ops()
    _fwd_case("halo128_ph8", "bf16", 2, 16, 64, 0, 128, 3, 1)
    x = torch.zeros((1, 12, 12, 64), device="cuda")
    w = torch.zeros(9 * 64 * 64, device="cuda")
    y = torch.zeros((1, 12, 12, 64), device="cuda")
    ops.set_tuning("tapgemm.variant", variant)
    with pytest.raises(ShmError):             # 12 x 12 is not a whole number of 16 x 16 patches
        ops.conv2d_fwd(x, None, 0, 64, 0, w, None, y, 64, 1, 12, 12, 64, 64, 3, 1, 1.0)
    with pytest.raises(ShmError):             # stride 2
        ops.conv2d_fwd(torch.zeros((1, 32, 32, 64), device="cuda"), None, 0, 64, 0, w, None, y, 64, 1, 32, 32, 64, 64, 3, 2, 1.0)
    with pytest.raises(ShmError):
        ops.set_tuning("tapgemm.variant", 99)
    with pytest.raises(ShmError):
        ops.set_tuning("no.such.knob", 1)
    ops.set_tuning("tapgemm.variant", -1)
    assert ops.get_tuning("tapgemm.variant") == 0 and ops.get_tuning("tapgemm.halo_min_blocks") == 1024


# ---- shapes only the DMA tap GEMM takes: stride 2, 1x1, maps that are not multiples of 16, tail rows
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant", DMA)
@pytest.mark.parametrize("n,h,c1,c2,cout,k,s", [
    (2, 32, 64, 0, 128, 3, 2),       # discriminator block
    (3, 10, 64, 0, 96, 3, 1),        # 10 x 10 map: M = 300 (tail rows in every tile height), ragged N
    (2, 16, 128, 0, 64, 1, 1),       # 1x1 bottleneck
    (1, 8, 64, 64, 192, 3, 1),       # concat, M = 64 < every tile height
])
def test_dma_forced_variant_other_shapes(variant, dt, n, h, c1, c2, cout, k, s):
    _fwd_case(variant, dt, n, h, c1, c2, cout, k, s, seed=1)


# ---- input gradient (flipped taps / four stride-2 phases, split destination) under forced variants
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant", HALO + ["dma128x128", "dma64x128", "dma128x64", "dma64x64", "dma256x64", "dma256x128", "wreg", "wreg_pp"])
def test_dgrad_s1_forced_variant(variant, dt):
    ops = _ops()
    rng = np.random.default_rng(7)
    n, h, c1, c2, cout = 2, 16, 64, 64, 64              # dx split into (upsampled, skip) parts: n1 = 64
    if variant == "wreg_pp":                             # a 32 x 32 map: the bf16 launch takes the ping-pong kernel
        variant, n, h = "wreg", 3, 32
    cin = c1 + c2
    w = rng.standard_normal((3, 3, cin, cout)) * 0.1
    dy = rng.standard_normal((n, h, h, cout))
    xt = torch.zeros(n, cin, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(xt, t64(_rnd(w, dt)), 1), xt, nchw(_rnd(dy, dt)))
    ref = nhwc(ref)
    adt = BF if dt == "bf16" else torch.float32
    d1 = torch.full((n, h, h, c1), 7.0, device="cuda", dtype=adt)
    d2 = torch.full((n, h, h, c2), 7.0, device="cuda", dtype=adt)
    ops.set_tuning("tapgemm.variant", variant)
    ops.conv2d_dgrad(_dev(dy, dt), cout, _dev(w, dt), d1, d2, c1, c1, c2, n, h, h, cin, cout, 3, 1)
    assert ops.last_kernel() == _sym(variant, dt, epi=False, hw=(h, h))
    if variant == "wreg" and dt == "bf16":    # the same product with the gradient signal leaving in fp32 (SHM_BF16_GF32)
        f1 = torch.full((n, h, h, c1), 7.0, device="cuda")
        f2 = torch.full((n, h, h, c2), 7.0, device="cuda")
        ops.conv2d_dgrad(_dev(dy, dt), cout, _dev(w, dt), f1, f2, c1, c1, c2, n, h, h, cin, cout, 3, 1)
        assert ops.last_kernel() == "tapgemm_wreg_kernel<float, 2>"
        assert rel_l2(host(f1), ref[..., :c1]) < 1e-4 and rel_l2(host(f2), ref[..., c1:]) < 1e-4
    assert rel_l2(host(d1.float()), ref[..., :c1]) < TOL[dt] and rel_l2(host(d2.float()), ref[..., c1:]) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant", ["dma128x128", "dma64x128", "dma128x64", "dma64x64", "dma256x64", "dma256x128", "dma128x128_bk32"])
def test_dgrad_s2_and_transpose_forced_variant(variant, dt):
    """Stride-2 input gradient and Conv2DTranspose forward: the four-phase launches (blockIdx.z = output phase)."""
    ops = _ops()
    rng = np.random.default_rng(8)
    n, h, cin, cout = 2, 16, 64, 128
    w = rng.standard_normal((3, 3, cin, cout)) * 0.1
    dy = rng.standard_normal((n, h // 2, h // 2, cout))
    xt = torch.zeros(n, cin, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(xt, t64(_rnd(w, dt)), 2), xt, nchw(_rnd(dy, dt)))
    adt = BF if dt == "bf16" else torch.float32
    dx = torch.full((n, h, h, cin), 7.0, device="cuda", dtype=adt)
    ops.set_tuning("tapgemm.variant", variant)
    ops.conv2d_dgrad(_dev(dy, dt), cout, _dev(w, dt), dx, None, cin, cin, 0, n, h, h, cin, cout, 3, 2)
    assert ops.last_kernel() == _sym(variant, dt)
    assert rel_l2(host(dx.float()), nhwc(ref)) < TOL[dt]
    # Conv2DTranspose(3x3, s2) + bias + LeakyReLU
    hi, ci, co = 8, 64, 96
    x = rng.standard_normal((n, hi, hi, ci))
    wt = rng.standard_normal((3, 3, co, ci)) * 0.1
    b = rng.standard_normal(co)
    r = nhwc(st.conv2d_transpose_same(nchw(_rnd(x, dt)), t64(_rnd(wt, dt)))) + b
    r = np.where(r > 0, r, 0.2 * r)
    y = torch.empty((n, 2 * hi, 2 * hi, co), device="cuda", dtype=adt)
    ops.conv2d_transpose_fwd(_dev(x, dt), ci, _dev(wt, dt), torch.from_numpy(b.astype(np.float32)).cuda(), y, co, n, hi, hi, ci, co, 0.2)
    assert ops.last_kernel() == _sym(variant, dt)
    assert rel_l2(host(y.float()), r) < TOL[dt]


# ---- the four phases fused in one block (tapgemm_phase4_kernel): 16 x 16 input patches, 64-channel output slices
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("n,h,cin,cout", [
    (2, 32, 64, 128),        # dy 16 x 16: one patch per image, K = 128
    (3, 64, 128, 64),        # four patches per image, two 64-channel output slices
    (1, 96, 192, 64),        # 3 x 3 patches (interior patch with all four neighbours), three output slices, one K chunk in bf16
])
def test_dgrad_s2_phase4(dt, n, h, cin, cout):
    ops = _ops()
    rng = np.random.default_rng(18)
    w = rng.standard_normal((3, 3, cin, cout)) * 0.1
    dy = rng.standard_normal((n, h // 2, h // 2, cout))
    xt = torch.zeros(n, cin, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(xt, t64(_rnd(w, dt)), 2), xt, nchw(_rnd(dy, dt)))
    adt = BF if dt == "bf16" else torch.float32
    dx = torch.full((n, h, h, cin), 7.0, device="cuda", dtype=adt)
    ops.set_tuning("tapgemm.variant", "phase4")
    ops.conv2d_dgrad(_dev(dy, dt), cout, _dev(w, dt), dx, None, cin, cin, 0, n, h, h, cin, cout, 3, 2)
    assert ops.last_kernel() == _sym("phase4", dt)
    assert rel_l2(host(dx.float()), nhwc(ref)) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("n,hi,ci,co", [(2, 16, 64, 64), (3, 32, 128, 192), (1, 48, 64, 128)])
def test_conv2d_transpose_phase4(dt, n, hi, ci, co):
    """Conv2DTranspose(3x3, s2) + bias + LeakyReLU through the fused four-phase kernel; shapes it cannot take are refused."""
    from shmgan_amd._lib import ShmError
    ops = _ops()
    rng = np.random.default_rng(19)
    x = rng.standard_normal((n, hi, hi, ci))
    wt = rng.standard_normal((3, 3, co, ci)) * 0.1
    b = rng.standard_normal(co)
    r = nhwc(st.conv2d_transpose_same(nchw(_rnd(x, dt)), t64(_rnd(wt, dt)))) + b
    r = np.where(r > 0, r, 0.2 * r)
    adt = BF if dt == "bf16" else torch.float32
    y = torch.full((n, 2 * hi, 2 * hi, co), 5.0, device="cuda", dtype=adt)
    ops.set_tuning("tapgemm.variant", "phase4")
    ops.conv2d_transpose_fwd(_dev(x, dt), ci, _dev(wt, dt), torch.from_numpy(b.astype(np.float32)).cuda(), y, co, n, hi, hi, ci, co, 0.2)
    assert ops.last_kernel() == _sym("phase4", dt)
    assert rel_l2(host(y.float()), r) < TOL[dt]
    y8 = torch.empty((1, 16, 16, 64), device="cuda", dtype=adt)
    with pytest.raises(ShmError):             # 8 x 8 input map: not a whole 16 x 16 patch
        ops.conv2d_transpose_fwd(torch.zeros((1, 8, 8, 64), device="cuda", dtype=adt), 64, torch.zeros((3, 3, 64, 64), device="cuda", dtype=adt), None,
                                 y8, 64, 1, 8, 8, 64, 64, 0.2)
    with pytest.raises(ShmError):             # Cout = 96 is not a whole number of 64-channel slices
        ops.conv2d_transpose_fwd(torch.zeros((1, 16, 16, 64), device="cuda", dtype=adt), 64, torch.zeros((3, 3, 96, 64), device="cuda", dtype=adt), None,
                                 torch.empty((1, 32, 32, 96), device="cuda", dtype=adt), 96, 1, 16, 16, 64, 96, 0.2)
    with pytest.raises(ShmError):             # a unit-stride layer has one phase
        ops.conv2d_fwd(torch.zeros((1, 16, 16, 64), device="cuda", dtype=adt), None, 0, 64, 0, torch.zeros(9 * 64 * 64, device="cuda", dtype=adt), None,
                       y8, 64, 1, 16, 16, 64, 64, 3, 1, 1.0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_default_dispatch_transpose_at_real_shape(dt):
    """Generator conv2d_transpose_3 (128 -> 64, 128 x 128 -> 256 x 256) at n = 8 through the default dispatch: 512 fused blocks."""
    ops = _ops()
    rng = np.random.default_rng(24)
    n, hi, ci, co = 8, 128, 128, 64
    x = rng.standard_normal((n, hi, hi, ci)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, co, ci)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(co) * 0.1).astype(np.float32)
    r = nhwc(st.conv2d_transpose_same(nchw(_rnd(x, dt)), t64(_rnd(wt, dt)))) + b
    r = np.where(r > 0, r, 0.2 * r)
    adt = BF if dt == "bf16" else torch.float32
    y = torch.empty((n, 2 * hi, 2 * hi, co), device="cuda", dtype=adt)
    ops.conv2d_transpose_fwd(_dev(x, dt), ci, _dev(wt, dt), torch.from_numpy(b).cuda(), y, co, n, hi, hi, ci, co, 0.2)
    assert ops.last_kernel() == _sym("phase4", dt), ops.last_kernel()
    assert rel_l2(host(y.float()), r) < TOL[dt]


# ---- weight gradient: generic / halo / thin-input kernels on the same shapes
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("wv,expect", [(0, None), (1, "wgrad_"), (2, "wgrad_halo")])
@pytest.mark.parametrize("n,h,cin,cout,blocks", [(2, 32, 64, 64, 0), (3, 16, 10, 64, 0), (2, 32, 128, 192, 64), (4, 16, 64, 128, 7)])
def test_wgrad_forced_variant(dt, wv, expect, n, h, cin, cout, blocks):
    ops = _ops()
    rng = np.random.default_rng(9)
    pitch = 32 if dt == "bf16" else 16
    ld = (cin + pitch - 1) // pitch * pitch
    x = np.zeros((n, h, h, ld))
    x[..., :cin] = rng.standard_normal((n, h, h, cin))
    dy = rng.standard_normal((n, h, h, cout))
    wt = torch.zeros(3, 3, cin, cout, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(nchw(_rnd(x[..., :cin], dt)), wt, 1), wt, nchw(_rnd(dy, dt)))
    ops.set_tuning("wgrad.variant", wv)
    ops.set_tuning("wgrad.blocks", blocks)
    ops.set_tuning("wgrad.bf16_wide", 4)          # the eight-wave bf16 block at unit stride too (the automatic choice takes it at stride 2 only)
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h, h, cin, cout, 3) // 4 + 1024, device="cuda")
    for rows in ((0, 2) if dt == "bf16" and wv != 1 else (0,)):        # bf16 halo kernel: 4 (automatic here) and 2 pixel rows per stage
        ops.set_tuning("wgrad.bf16_rows", rows)
        dw = torch.full((3, 3, cin, cout), 3.0, device="cuda")
        ops.conv2d_wgrad(_dev(x, dt), None, 0, ld, 0, _dev(dy, dt), cout, dw, n, h, h, cin, ld, cout, 3, 1, 0, ws)
        k = ops.last_kernel()
        if wv == 1:
            assert k.startswith("wgrad_kernel<") or k.startswith("wgrad_bf16_kernel<"), k
        elif wv == 2:
            # bf16 with >= 128 output channels on a map whose height is a multiple of four: the eight-wave 64 x 128 block (round 4)
            wide = dt == "bf16" and cout >= 128 and rows == 0
            assert k == ("wgrad_halo8_bf16_kernel<0>" if wide else "wgrad_halo_kernel" if dt == "f32" else f"wgrad_halo_bf16_kernel<{rows or 4}>"), k
        assert rel_l2(host(dw), ref.numpy()) < (1e-4 if dt == "bf16" else 1e-5), (k, rel_l2(host(dw), ref.numpy()))
        if dt == "bf16" and wv == 2 and rows == 0 and cout >= 128:       # ... and the four-wave kernel on the same shape ("wgrad.bf16_wide" = 1)
            ops.set_tuning("wgrad.bf16_wide", 1)
            dw4 = torch.full((3, 3, cin, cout), 3.0, device="cuda")
            ops.conv2d_wgrad(_dev(x, dt), None, 0, ld, 0, _dev(dy, dt), cout, dw4, n, h, h, cin, ld, cout, 3, 1, 0, ws)
            assert ops.last_kernel() == "wgrad_halo_bf16_kernel<4>", ops.last_kernel()
            assert rel_l2(host(dw4), ref.numpy()) < 1e-4
            ops.set_tuning("wgrad.bf16_wide", 4)


# ---- stride-2 3x3 weight gradient: halo form (2 x 8 output patches, 5 x 17 input halo) against the generic kernel and float64
@pytest.mark.parametrize("n,h,cin,cout,c1,blocks", [
    (2, 32, 64, 128, 0, 0), (3, 16, 64, 64, 0, 0), (1, 64, 128, 64, 0, 0), (2, 32, 96, 80, 0, 0), (2, 32, 128, 64, 64, 0),
    (2, 32, 64, 128, 0, 7), (5, 16, 192, 64, 128, 3), (1, 16, 32, 16, 0, 0),
])
def test_wgrad_stride2_halo(n, h, cin, cout, c1, blocks):
    ops = _ops()
    rng = np.random.default_rng(31)
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    dy = rng.standard_normal((n, h // 2, h // 2, cout)).astype(np.float32)
    wt = torch.zeros(3, 3, cin, cout, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(nchw(x.astype(np.float64)), wt, 2), wt, nchw(dy.astype(np.float64)))
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, cin, cout, 3) // 4 + 1024, device="cuda")
    xa = torch.from_numpy(np.ascontiguousarray(x[..., :c1] if c1 else x)).cuda()
    xb = torch.from_numpy(np.ascontiguousarray(x[..., c1:])).cuda() if c1 else None
    dyd = torch.from_numpy(dy).cuda()
    got = {}
    for wv in (0, 3):
        ops.set_tuning("wgrad.variant", wv)
        ops.set_tuning("wgrad.blocks", blocks)
        dw = torch.full((3, 3, cin, cout), 3.0, device="cuda")
        ops.conv2d_wgrad(xa, xb, c1, c1 if c1 else cin, cin - c1 if c1 else 0, dyd, cout, dw, n, h, h, cin, cin, cout, 3, 2, 0, ws)
        k = ops.last_kernel()
        assert k == ("wgrad_halo_kernel<0, true>" if wv == 0 else "wgrad_kernel<9, false>"), k
        got[wv] = host(dw)
        assert rel_l2(got[wv], ref.numpy()) < 1e-5, (k, rel_l2(got[wv], ref.numpy()))
        # accumulate into the result
        ops.conv2d_wgrad(xa, xb, c1, c1 if c1 else cin, cin - c1 if c1 else 0, dyd, cout, dw, n, h, h, cin, cin, cout, 3, 2, 1, ws)
        assert rel_l2(host(dw), 2 * ref.numpy()) < 1e-5
    ops.set_tuning("reset", 0)
    assert rel_l2(got[0], got[3]) < 1e-5


# ---- bf16 stride-2 3x3 weight gradient (round 4): the eight-wave block on 2 x 16 output patches with the even / odd column runs of a
# 5 x 33 input halo, against float64 on the bf16-rounded operands and against the generic kernel it replaces (wgrad_bf16_kernel<9>)
@pytest.mark.parametrize("n,h,cin,cout,c1,blocks", [
    (2, 32, 64, 128, 0, 0),        # one patch column, one 64 x 128 tile
    (1, 64, 128, 256, 0, 0),       # two patch columns, 2 x 2 tiles
    (2, 64, 128, 128, 64, 0),      # Concatenate split on a tile boundary
    (3, 32, 96, 192, 0, 5),        # ragged ci (96) and co (192 = 128 + 64) tiles, odd batch, odd split count
    (5, 32, 64, 128, 0, 3),        # blocks that cross image boundaries
    (3, 16, 128, 256, 0, 0),       # 8-column output map: stages of 4 x 8 output pixels (wgrad_halo8_bf16_kernel<2>), one patch per image
    (2, 48, 64, 128, 0, 5),        # 24 x 24 outputs: three 8-column patch columns, six patch rows
])
def test_wgrad_stride2_bf16_wide(n, h, cin, cout, c1, blocks):
    ops = _ops()
    rng = np.random.default_rng(41)
    x = rng.standard_normal((n, h, h, cin))
    dy = rng.standard_normal((n, h // 2, h // 2, cout))
    wt = torch.zeros(3, 3, cin, cout, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(nchw(_rnd(x, "bf16")), wt, 2), wt, nchw(_rnd(dy, "bf16")))
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, cin, cout, 3) // 4 + 1024, device="cuda")
    xa = _dev(np.ascontiguousarray(x[..., :c1] if c1 else x), "bf16")
    xb = _dev(np.ascontiguousarray(x[..., c1:]), "bf16") if c1 else None
    dyd = _dev(dy, "bf16")
    got = {}
    for wide in (0, 1):
        ops.set_tuning("wgrad.bf16_wide", wide)
        ops.set_tuning("wgrad.blocks", blocks)
        ws.fill_(float("nan"))                    # every slab element a block owns must be written
        dw = torch.full((3, 3, cin, cout), 3.0, device="cuda")
        ops.conv2d_wgrad(xa, xb, c1, c1 if c1 else cin, cin - c1 if c1 else 0, dyd, cout, dw, n, h, h, cin, cin, cout, 3, 2, 0, ws)
        k = ops.last_kernel()
        assert k == (f"wgrad_halo8_bf16_kernel<{1 if (h // 2) % 16 == 0 else 2}>" if wide == 0 else "wgrad_bf16_kernel<9, false>"), k
        got[wide] = host(dw)
        assert rel_l2(got[wide], ref.numpy()) < 1e-4, (k, rel_l2(got[wide], ref.numpy()))
        ops.conv2d_wgrad(xa, xb, c1, c1 if c1 else cin, cin - c1 if c1 else 0, dyd, cout, dw, n, h, h, cin, cin, cout, 3, 2, 1, ws)     # accumulate
        assert rel_l2(host(dw), 2 * ref.numpy()) < 1e-4
    ops.set_tuning("reset", 0)
    assert rel_l2(got[0], got[1]) < 1e-5          # both accumulate in fp32 from the same bf16 products: only the summation order differs


def test_wgrad_stride2_bf16_wide_leaves_what_it_cannot_tile():
    """a 4-column output map, 64 output channels and "wgrad.variant" 3 stay on the generic bf16 kernel"""
    ops = _ops()
    rng = np.random.default_rng(42)
    for n, h, cin, cout, wv in ((2, 8, 64, 128, 0), (2, 32, 64, 64, 0), (2, 32, 64, 128, 3)):
        x = rng.standard_normal((n, h, h, cin))
        dy = rng.standard_normal((n, h // 2, h // 2, cout))
        wt = torch.zeros(3, 3, cin, cout, dtype=torch.float64, requires_grad=True)
        ref, = torch.autograd.grad(st.conv2d_same(nchw(_rnd(x, "bf16")), wt, 2), wt, nchw(_rnd(dy, "bf16")))
        ws = torch.empty(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, cin, cout, 3) // 4 + 1024, device="cuda")
        dw = torch.empty((3, 3, cin, cout), device="cuda")
        ops.set_tuning("wgrad.variant", wv)
        ops.conv2d_wgrad(_dev(x, "bf16"), None, 0, cin, 0, _dev(dy, "bf16"), cout, dw, n, h, h, cin, cin, cout, 3, 2, 0, ws)
        assert ops.last_kernel() == "wgrad_bf16_kernel<9, false>", ops.last_kernel()
        assert rel_l2(host(dw), ref.numpy()) < 1e-4


def test_wgrad_stride2_halo_refuses_what_it_cannot_tile():
    """an 8 x 8 map (4 output columns) and a concat split inside a 64-channel tile stay on the generic kernel"""
    ops = _ops()
    rng = np.random.default_rng(32)
    for n, h, cin, cout, c1 in ((2, 8, 64, 64, 0), (2, 32, 96, 64, 32)):
        x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
        dy = rng.standard_normal((n, h // 2, h // 2, cout)).astype(np.float32)
        wt = torch.zeros(3, 3, cin, cout, dtype=torch.float64, requires_grad=True)
        ref, = torch.autograd.grad(st.conv2d_same(nchw(x.astype(np.float64)), wt, 2), wt, nchw(dy.astype(np.float64)))
        ws = torch.empty(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, cin, cout, 3) // 4 + 1024, device="cuda")
        xa = torch.from_numpy(np.ascontiguousarray(x[..., :c1] if c1 else x)).cuda()
        xb = torch.from_numpy(np.ascontiguousarray(x[..., c1:])).cuda() if c1 else None
        dw = torch.empty((3, 3, cin, cout), device="cuda")
        ops.conv2d_wgrad(xa, xb, c1, c1 if c1 else cin, cin - c1 if c1 else 0, torch.from_numpy(dy).cuda(), cout, dw, n, h, h, cin, cin, cout, 3, 2, 0, ws)
        assert ops.last_kernel().startswith("wgrad_kernel<9"), ops.last_kernel()
        assert rel_l2(host(dw), ref.numpy()) < 1e-5


# ======================================================================================================
# Real layer shapes of BASELINE configs[1] (S=256, F=64, B=8) through the DEFAULT dispatch.  The float64
# reference convolutions take a few seconds each on the GPU box's host cores.
# ======================================================================================================
def test_default_dispatch_fp32_halo128_on_a_generator_layer():
    """Generator conv2d_5 (128 -> 128 at 128 x 128) through the default dispatch: n = 6 is 384 128-wide halo blocks on 256 CUs
    (the busiest CU gets two) against 768 64-wide ones (three halves) -> the 64-wide static-tap halo block; n = 8 (the G(1) batch,
    two per CU) and n = 16 -> the 128-wide one."""
    ops = _ops()
    rng = np.random.default_rng(21)
    h, cin, cout = 128, 128, 128
    w = rng.standard_normal((3, 3, cin, cout)) * 0.05
    b = rng.standard_normal(cout) * 0.1
    wk = _wk(w, cin, "f32")
    for n, sym in ((6, "halo64_st"), (8, "halo128_st"), (16, "halo128_st")):
        x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
        ref = conv_ref(x, w.astype(np.float32), 1) + b
        ref = np.where(ref > 0, ref, 0.2 * ref)
        y = torch.empty((n, h, h, cout), device="cuda")
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
        ops.conv2d_in_fwd(_dev(x, "f32"), None, 0, cin, 0, wk, torch.from_numpy(b.astype(np.float32)).cuda(), y, cout, n, h, h, cin, cout,
                          3, 1, 0.2, stats, 1e-6, scratch=scr)
        assert ops.last_kernel() == _sym(sym, "f32"), ops.last_kernel()
        got = host(y)
        assert rel_l2(got, ref) < 1e-5
        _check_stats(stats, got, n, cout, "f32")


def test_default_dispatch_fp32_dgrad_256_and_dma128x128():
    """dgrad of generator conv2d_24 at full resolution (n = 8, 256 x 256, 128 <- 64, the gradient split into its upsampled and
    skip halves: K = 64 -> the weights-in-registers kernel, two N tiles), a 128 <- 128 dgrad at n = 16 (static-tap halo 128) and a
    discriminator block (64 -> 128, stride 2, n = 32: K = 64 and 512 tiles of 256 x 128 -> the eight-wave DMA tile; the same block
    with 256 input channels -> DMA 128x128)."""
    ops = _ops()
    rng = np.random.default_rng(22)
    n, h, cin, cout = 8, 256, 128, 64
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    dy = rng.standard_normal((n, h, h, cout)).astype(np.float32)
    xt = torch.zeros(n, cin, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(xt, t64(w), 1), xt, nchw(dy))
    d1 = torch.empty((n, h, h, 64), device="cuda")
    d2 = torch.empty((n, h, h, 64), device="cuda")
    ops.conv2d_dgrad(_dev(dy, "f32"), cout, _dev(w, "f32"), d1, d2, 64, 64, 64, n, h, h, cin, cout, 3, 1)
    assert ops.last_kernel() == _sym("wreg", "f32"), ops.last_kernel()
    ref = nhwc(ref.detach())
    assert rel_l2(host(d1), ref[..., :64]) < 1e-5 and rel_l2(host(d2), ref[..., 64:]) < 1e-5
    del ref, xt
    n, h, c = 16, 128, 128                               # 128 <- 128 at 128 x 128: 1024 halo blocks
    w = (rng.standard_normal((3, 3, c, c)) * 0.05).astype(np.float32)
    dy = rng.standard_normal((n, h, h, c)).astype(np.float32)
    xt = torch.zeros(n, c, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(xt, t64(w), 1), xt, nchw(dy))
    dx = torch.empty((n, h, h, c), device="cuda")
    ops.conv2d_dgrad(_dev(dy, "f32"), c, _dev(w, "f32"), dx, None, c, c, 0, n, h, h, c, c, 3, 1)
    assert ops.last_kernel() == _sym("halo128_st", "f32"), ops.last_kernel()
    assert rel_l2(host(dx), nhwc(ref.detach())) < 1e-5
    del ref, xt
    n, h, cin, cout = 32, 128, 64, 128
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    ref = conv_ref(x, w, 2)
    ref = np.where(ref > 0, ref, 0.2 * ref)
    y = torch.empty((n, h // 2, h // 2, cout), device="cuda")
    ops.conv2d_fwd(_dev(x, "f32"), None, 0, cin, 0, _wk(w, cin, "f32"), None, y, cout, n, h, h, cin, cout, 3, 2, 0.2)
    assert ops.last_kernel() == _sym("dma256x128", "f32"), ops.last_kernel()
    assert rel_l2(host(y), ref) < 1e-5
    n, h, cin, cout = 16, 64, 256, 256
    x = rng.standard_normal((n, h, h, cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    ref = conv_ref(x, w, 2)
    ref = np.where(ref > 0, ref, 0.2 * ref)
    for dt, sym in (("f32", "dma64x128"), ("bf16", "dma128x128_bk32")):       # 256 tiles of 128 x 128: fp32 small-grid tile, bf16 long-K tile
        y = torch.empty((n, h // 2, h // 2, cout), device="cuda", dtype=BF if dt == "bf16" else torch.float32)
        ops.conv2d_fwd(_dev(x, dt), None, 0, cin, 0, _wk(w, cin, dt), None, y, cout, n, h, h, cin, cout, 3, 2, 0.2)
        assert ops.last_kernel() == _sym(sym, dt), ops.last_kernel()
        r = conv_ref(_rnd(x, dt), _rnd(w, dt), 2)
        assert rel_l2(host(y.float()), np.where(r > 0, r, 0.2 * r)) < TOL[dt]


def test_default_dispatch_fp32_grid_smaller_than_the_chip():
    """SpecSeg's deep layers at n = 8 (and any model at batch 1): 128 -> 128 at 32 x 32 is 64 64-wide halo blocks for 256 CUs -> the
    64 x 64 DMA tile (256 blocks), forward with fused statistics and input gradient; the same layer in bf16 stays on the halo block."""
    ops = _ops()
    rng = np.random.default_rng(25)
    n, h, c = 8, 32, 128
    x = rng.standard_normal((n, h, h, c)).astype(np.float32)
    w = (rng.standard_normal((3, 3, c, c)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(c) * 0.1).astype(np.float32)
    for dt, sym in (("f32", "dma64x64"), ("bf16", "halo64_st")):
        ref = conv_ref(_rnd(x, dt), _rnd(w, dt), 1) + b
        ref = np.where(ref > 0, ref, 0.2 * ref)
        adt = BF if dt == "bf16" else torch.float32
        y = torch.empty((n, h, h, c), device="cuda", dtype=adt)
        stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
        ops.conv2d_in_fwd(_dev(x, dt), None, 0, c, 0, _wk(w, c, dt), torch.from_numpy(b).cuda(), y, c, n, h, h, c, c, 3, 1, 0.2, stats, 1e-6, scratch=scr)
        assert ops.last_kernel() == _sym(sym, dt), ops.last_kernel()
        got = host(y.float())
        assert rel_l2(got, ref) < TOL[dt]
        _check_stats(stats, got, n, c, dt)
        assert float(scr.abs().max()) == 0.0
    dy = rng.standard_normal((n, h, h, c)).astype(np.float32)
    xt = torch.zeros(n, c, h, h, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad(st.conv2d_same(xt, t64(w), 1), xt, nchw(dy))
    dx = torch.empty((n, h, h, c), device="cuda")
    ops.conv2d_dgrad(_dev(dy, "f32"), c, _dev(w, "f32"), dx, None, c, c, 0, n, h, h, c, c, 3, 1)
    assert ops.last_kernel() == _sym("dma64x64", "f32"), ops.last_kernel()
    assert rel_l2(host(dx), nhwc(ref.detach())) < 1e-5


def test_default_dispatch_fp32_cout64_at_256():
    """The headline block: 64 -> 64 at 256 x 256, n = 8 (fp32 default = the weights-in-registers kernel)."""
    ops = _ops()
    rng = np.random.default_rng(23)
    n, h, c = 8, 256, 64
    x = rng.standard_normal((n, h, h, c)).astype(np.float32)
    w = (rng.standard_normal((3, 3, c, c)) * 0.05).astype(np.float32)
    b = (rng.standard_normal(c) * 0.1).astype(np.float32)
    ref = conv_ref(x, w, 1) + b
    ref = np.where(ref > 0, ref, 0.2 * ref)
    y = torch.empty((n, h, h, c), device="cuda")
    stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
    ops.conv2d_in_fwd(_dev(x, "f32"), None, 0, c, 0, _wk(w, c, "f32"), torch.from_numpy(b).cuda(), y, c, n, h, h, c, c, 3, 1, 0.2, stats,
                      1e-6, scratch=scr)
    assert ops.last_kernel() == _sym("wreg", "f32"), ops.last_kernel()
    got = host(y)
    assert rel_l2(got, ref) < 1e-5
    _check_stats(stats, got, n, c, "f32")
