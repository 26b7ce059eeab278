"""The fused block's forward (round 3): InstanceNormalization applied by the CONSUMER (SHM.py:244-245, Conv -> LeakyReLU -> IN;
include/shmgan_hip.h: shm_conv2d_in_fwd_norm / shm_conv2d_wgrad_norm / shm_in_pool / *_norm_supported).

The kernels that stage their A operand as an LDS halo image normalise it there, from the producing block's (mean, inv, beta)
table, so the normalised tensor is never written.  The arithmetic is shm_in_apply's (common.h: shm_in_norm) and out-of-image
taps stay zero, hence every check here is BITWISE against the two-pass path (shm_in_apply, then the plain entry point):
  * every folding forward kernel, forced through shm_set_tuning, one source and the Concatenate form (second source folded),
    maps with 1..4 patches per side (every border case of the 3x3 halo), several samples (the per-sample table);
  * both weight-gradient kernels (fp32, bf16 with 2 and 4 pixel rows per stage);
  * shm_in_pool against shm_in_apply_pool's pooled output, shm_in_norm_table against the by-product of shm_conv2d_in_fwd_norm;
  * shapes the chosen kernel cannot fold are refused (SHM_E_SHAPE), never silently mis-computed, and the *_supported queries
    agree with what the entry points then do;
  * the whole train_step with the fold on and off: same losses and gradients to the run-to-run bound of a step (the statistics
    sums are float64 atomics, whose order is not fixed: tests/test_train_loop_gpu.py::test_step_is_reproducible_run_to_run).
The second half of the file is SHM_NORM_SCALED, the same fold carried by the operands (per-sample weights w * inv and bias rows, `ring`
in the out-of-image taps, weight-gradient slabs scaled per sample + a rank-n term): equal to the two-pass path to ROUNDING -- fp32
outputs 3e-6, weight gradients 2e-5, border rows and columns on their own, the whole step within the oracle contract -- plus the
per-sample dz sums the InstanceNorm backward keeps for it and the pixel mappings of the backward passes ("elem.interleave").
"""
import numpy as np
import pytest
import torch

from oracle import step_torch as st
from util import host, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
EPS = 1e-6


def _ops():
    from shmgan_amd import ops
    return ops


@pytest.fixture(autouse=True)
def _reset_tuning():
    # the bit-for-bit comparisons of this module are between the folding form and the plain form of ONE kernel family.  In bf16 the
    # plain weights-in-registers launch defaults to tapgemm_wreg16_bf16_kernel (round 4: halo-row walk, another fp32 summation order
    # than the four-wave kernel the folding forms live in -- equal to rounding, not to the bit): the plain calls are pinned to that kernel
    _ops().set_tuning("tapgemm.wreg16", 0)
    yield
    _ops().set_tuning("reset", 0)


def _t(a, dt):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    return t.to(BF) if dt == "bf16" else t


def _adt(dt):
    return BF if dt == "bf16" else torch.float32


def _block(rng, n, h, c, dt):
    """An un-normalised activation, its finalized statistics, beta, its table and its normalised tensor (shm_in_apply)."""
    ops = _ops()
    a = _t(rng.standard_normal((n, h, h, c)) * rng.uniform(0.5, 2.0, (n, 1, 1, c)) + rng.uniform(-1, 1, (n, 1, 1, c)), dt)
    beta = _t(rng.uniform(-0.5, 0.5, c), "f32")
    stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, EPS)
    nt = torch.full((n, 4, c), 9.0, dtype=torch.float32, device="cuda")
    ops.in_norm_table(stats, beta, nt, n, c)
    ahat = torch.empty_like(a)
    ops.in_apply(a, c, stats, beta, ahat, c, n, h * h, c)
    return a, stats, beta, nt, ahat


def _wk(rng, cin, cout, dt):
    """K-contiguous weights [tap][cout][cin] as the forward kernels read them."""
    return _t(rng.standard_normal((9, cout, cin)) * 0.1, dt)


def _same_stats(s0, s1):
    """statistics of two launches over identical outputs: per-lane fp32 partial sums (sum v * v is an fma in one instantiation of a
    kernel and a multiply-add in another), float64 atomics across lanes: equal to fp32 rounding, not to the bit"""
    return torch.allclose(s0, s1, rtol=1e-6, atol=1e-7)


def _in_fwd(x, x2, c1, wk, bias, n, h, cin, cout, dt, **kw):
    ops = _ops()
    y = torch.full((n, h, h, cout), 7.0, device="cuda", dtype=_adt(dt))
    stats = torch.zeros(n * cout * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
    ldx = x.shape[-1]
    ldx2 = 0 if x2 is None else x2.shape[-1]
    ops.conv2d_in_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, EPS, scratch=scr, **kw)
    torch.cuda.synchronize()
    assert float(scr.abs().max()) == 0.0
    return y, stats, ops.last_kernel()


FWD_CASES = [
    # variant, n, h, cin (c1 = 0: one source), cout
    ("wreg", 3, 16, 64, 64),
    ("wreg", 2, 48, 64, 128),
    ("wreg", 2, 32, 32, 64),
    ("halo64_st", 3, 16, 64, 64),
    ("halo64_st", 2, 32, 128, 64),
    ("halo128_st", 2, 16, 128, 128),
    ("halo128_st", 3, 32, 256, 128),
    ("halo128_st", 1, 64, 128, 256),
]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant,n,h,cin,cout", FWD_CASES)
def test_forward_fold_is_bit_identical(variant, n, h, cin, cout, dt):
    ops = _ops()
    if variant == "wreg" and dt == "f32" and cin == 32 and cout != 64:
        pytest.skip("fp32 wreg: 64 output channels per block")
    rng = np.random.default_rng(11)
    a, stats, beta, nt, ahat = _block(rng, n, h, cin, dt)
    wk, bias = _wk(rng, cin, cout, dt), _t(rng.standard_normal(cout) * 0.1, "f32")
    ops.set_tuning("tapgemm.variant", variant)
    y0, s0, k0 = _in_fwd(ahat, None, 0, wk, bias, n, h, cin, cout, dt)
    nt_out = torch.full((n, 4, cout), 5.0, dtype=torch.float32, device="cuda")
    beta_out = _t(rng.uniform(-1, 1, cout), "f32")
    y1, s1, k1 = _in_fwd(a, None, 0, wk, bias, n, h, cin, cout, dt, nt_x=nt, nt_out=nt_out, beta_out=beta_out)
    assert k1 != k0 and k1.rstrip(">").endswith(", 1"), (k0, k1)            # the SHM_NORM_EXACT instantiation ran
    assert torch.equal(y0, y1), (k1, float((y0.float() - y1.float()).abs().max()))
    assert _same_stats(s0, s1)
    # this block's own table, a by-product of the statistics finalisation, equals shm_in_norm_table's
    ref = torch.empty_like(nt_out)
    ops.in_norm_table(s1, beta_out, ref, n, cout)
    assert torch.equal(ref, nt_out)
    st64 = s1.view(n, cout, 2)
    assert torch.equal(nt_out[:, 0], st64[..., 0].float()) and torch.equal(nt_out[:, 1], st64[..., 1].float())
    assert torch.equal(nt_out[:, 2], beta_out.expand(n, cout))
    # ring = the raw value whose normalised image is 0
    ring = nt_out[:, 3].double()
    assert float(((ring - nt_out[:, 0].double()) * nt_out[:, 1].double() + nt_out[:, 2].double()).abs().max()) < 1e-5


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant,n,h,cu,cs,cout", [
    ("halo64_st", 2, 32, 64, 64, 64),            # the generator's top decoder level: [u, skip] -> 64
    ("halo128_st", 2, 16, 128, 128, 128),
    ("halo128_st", 1, 32, 256, 256, 256),        # 256 folded channels: the LDS table's limit
])
def test_concat_fold_second_source(variant, n, h, cu, cs, cout, dt):
    """Concatenate([u, skip]): u (a Conv2DTranspose output) is used as stored, the skip is normalised in LDS."""
    ops = _ops()
    rng = np.random.default_rng(12)
    u = _t(rng.standard_normal((n, h, h, cu)), dt)
    a, stats, beta, nt, ahat = _block(rng, n, h, cs, dt)
    cin = cu + cs
    wk, bias = _wk(rng, cin, cout, dt), _t(rng.standard_normal(cout) * 0.1, "f32")
    ops.set_tuning("tapgemm.variant", variant)
    y0, s0, _ = _in_fwd(u, ahat, cu, wk, bias, n, h, cin, cout, dt)
    y1, s1, k1 = _in_fwd(u, a, cu, wk, bias, n, h, cin, cout, dt, nt_x2=nt)
    assert torch.equal(y0, y1) and _same_stats(s0, s1), k1
    # ... and the first source folded instead (not a generator case, same code path with part 0)
    y2, s2, _ = _in_fwd(ahat, u, cs, wk, bias, n, h, cin, cout, dt)
    y3, s3, k3 = _in_fwd(a, u, cs, wk, bias, n, h, cin, cout, dt, nt_x=nt)
    assert torch.equal(y2, y3) and _same_stats(s2, s3), k3


def _wgrad(x, x2, c1, dy, n, h, cin, cout, **kw):
    ops = _ops()
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h, h, cin, cout, 3) // 4 + 16, device="cuda")
    dw = torch.full((3, 3, cin, cout), 3.0, device="cuda")
    ldx = x.shape[-1]
    ldx2 = 0 if x2 is None else x2.shape[-1]
    ops.conv2d_wgrad(x, x2, c1, ldx, ldx2, dy, cout, dw, n, h, h, cin, cin, cout, 3, 1, 0, ws, **kw)
    torch.cuda.synchronize()
    return dw, ops.last_kernel()


@pytest.mark.parametrize("dt,rows", [("f32", 0), ("bf16", 4), ("bf16", 2), ("f32x3", 2)])       # f32x3: "wgrad.f32_split" (csrc/conv_wgrad_x3.hip)
@pytest.mark.parametrize("n,h,cin,cout,blocks", [
    (3, 16, 64, 64, 0),
    (2, 32, 128, 64, 0),
    (5, 16, 64, 128, 7),              # few slabs: a block walks several images (the lane's table registers are re-read)
    (2, 48, 64, 64, 0),
])
def test_wgrad_fold_is_bit_identical(dt, rows, n, h, cin, cout, blocks):
    ops = _ops()
    rng = np.random.default_rng(13)
    x3 = dt == "f32x3"
    if x3:
        dt = "f32"
        ops.set_tuning("wgrad.f32_split", 1)
    a, stats, beta, nt, ahat = _block(rng, n, h, cin, dt)
    dy = _t(rng.standard_normal((n, h, h, cout)), dt)
    if rows:
        ops.set_tuning("wgrad.bf16_rows", rows)
    if blocks:
        ops.set_tuning("wgrad.blocks", blocks)
    dw0, k0 = _wgrad(ahat, None, 0, dy, n, h, cin, cout)
    dw1, k1 = _wgrad(a, None, 0, dy, n, h, cin, cout, nt_x=nt)
    if x3:
        assert (k0, k1) == ("wgrad_halo_x3_kernel<2>", "wgrad_halo_x3_kernel<2, true>"), (k0, k1)
    else:
        assert k1 != k0 and k1.rstrip(">").endswith("1"), (k0, k1)
    assert torch.equal(dw0, dw1), (k1, float((dw0 - dw1).abs().max()))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_wgrad_fold_concat(dt):
    ops = _ops()
    rng = np.random.default_rng(14)
    n, h, cu, cs, cout = 2, 32, 64, 128, 64
    u = _t(rng.standard_normal((n, h, h, cu)), dt)
    a, stats, beta, nt, ahat = _block(rng, n, h, cs, dt)
    dy = _t(rng.standard_normal((n, h, h, cout)), dt)
    dw0, _ = _wgrad(u, ahat, cu, dy, n, h, cu + cs, cout)
    dw1, k1 = _wgrad(u, a, cu, dy, n, h, cu + cs, cout, nt_x2=nt)
    assert torch.equal(dw0, dw1), k1


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_in_pool_matches_in_apply_pool(dt):
    ops = _ops()
    rng = np.random.default_rng(15)
    n, h, c = 3, 32, 64
    a, stats, beta, nt, ahat = _block(rng, n, h, c, dt)
    out = torch.empty_like(a)
    p0 = torch.empty((n, h // 2, h // 2, c), device="cuda", dtype=a.dtype)
    p1 = torch.full_like(p0, 3.0)
    ops.in_apply_pool(a, c, stats, beta, out, c, p0, c, n, h, h, c)
    ops.in_pool(a, c, stats, beta, p1, c, n, h, h, c)
    torch.cuda.synchronize()
    assert torch.equal(out, ahat) and torch.equal(p0, p1)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_unsupported_shapes_are_refused(dt):
    """A kernel that cannot normalise in LDS is never given an un-normalised source: the query says no and the call fails."""
    from shmgan_amd._lib import ShmError
    ops = _ops()
    rng = np.random.default_rng(16)
    adt = _adt(dt)
    n, h = 2, 16
    # (a) a forced DMA tile; (b) stride 2; (c) more folded channels than the LDS table holds
    for variant, cin, cout, stride in (("dma128x128", 64, 128, 1), ("auto", 64, 128, 2), ("halo128_st", 512, 128, 1)):
        ops.set_tuning("tapgemm.variant", variant)
        assert not ops.conv2d_norm_supported(n, h, h, cin, 0, cout, 3, stride, 0, adt), (variant, cin)
        a, stats, beta, nt, ahat = _block(rng, n, h, cin, dt)
        wk = _wk(rng, cin, cout, dt)
        y = torch.empty((n, h // stride, h // stride, cout), device="cuda", dtype=adt)
        s = torch.zeros(n * cout * 2, dtype=torch.float64, device="cuda")
        with pytest.raises(ShmError, match="cannot normalise"):
            ops.conv2d_in_fwd(a, None, 0, cin, 0, wk, None, y, cout, n, h, h, cin, cout, 3, stride, 0.2, s, EPS, nt_x=nt)
    ops.set_tuning("reset", 0)
    # queries follow the automatic choice: the generator's big layers fold, a map that is not a multiple of 16 does not
    assert ops.conv2d_norm_supported(8, 256, 256, 64, 0, 64, 3, 1, 0, adt)
    assert ops.conv2d_norm_supported(8, 128, 128, 256, 128, 128, 3, 1, 1, adt)
    assert not ops.conv2d_norm_supported(8, 24, 24, 64, 0, 64, 3, 1, 0, adt)
    assert ops.conv2d_wgrad_norm_supported(8, 256, 256, 64, 64, 0, 64, 3, 1, 0, adt)
    assert not ops.conv2d_wgrad_norm_supported(8, 256, 256, 64, 64, 0, 64, 3, 2, 0, adt)
    assert not ops.conv2d_wgrad_norm_supported(8, 128, 128, 192, 192, 96, 64, 3, 1, 1, adt)      # a 64-channel block would straddle the sources
    # wgrad refuses as well
    a, stats, beta, nt, ahat = _block(rng, n, h, 64, dt)
    dy = _t(rng.standard_normal((n, h // 2, h // 2, 64)), dt)
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, 64, 64, 3) // 4 + 16, device="cuda")
    dw = torch.zeros((3, 3, 64, 64), device="cuda")
    with pytest.raises(ShmError, match="cannot normalise"):
        ops.conv2d_wgrad(a, None, 0, 64, 0, dy, 64, dw, n, h, h, 64, 64, 64, 3, 2, 0, ws, nt_x=nt)


# ----------------------------------------------------------------------------- SHM_NORM_SCALED: the normalisation in the operands
def _prepare(wk, bias, nt, c, part_lo, n, cin, cout):
    ops = _ops()
    wk_n = torch.empty((n,) + tuple(wk.shape), device="cuda", dtype=wk.dtype)
    bias_n = torch.empty((n, cout), device="cuda")
    ops.conv2d_norm_prepare(wk, bias, nt, c, part_lo, wk_n, bias_n, n, cin, cout, 3)
    return wk_n, bias_n


def _close(got, ref, dt):
    """rel-L2 and worst element (relative to the tensor's scale): fp32 rounding of a re-associated sum / bf16 operand rounding"""
    g, r = got.double(), ref.double()
    rel = float((g - r).norm() / r.norm())
    worst = float((g - r).abs().max() / r.abs().max())
    return (rel < 3e-6 and worst < 3e-5) if dt == "f32" else (rel < 1.5e-2 and worst < 8e-2), (rel, worst)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant,n,h,cin,cout", FWD_CASES)
def test_forward_scaled_mode_matches_the_two_pass_path(variant, n, h, cin, cout, dt):
    """conv(w, (a - mean) * inv + beta) = conv(w * inv, a_ext) + bias_n with a_ext = `ring` outside the image: per-sample weights and bias
    rows (shm_conv2d_norm_prepare), the kernels only fill the out-of-image halo entries -- border pixels are where it could go wrong."""
    ops = _ops()
    if variant == "wreg" and dt == "f32" and cin == 32 and cout != 64:
        pytest.skip("fp32 wreg: 64 output channels per block")
    rng = np.random.default_rng(21)
    a, stats, beta, nt, ahat = _block(rng, n, h, cin, dt)
    wk, bias = _wk(rng, cin, cout, dt), _t(rng.standard_normal(cout) * 0.1, "f32")
    ops.set_tuning("tapgemm.variant", variant)
    y0, s0, k0 = _in_fwd(ahat, None, 0, wk, bias, n, h, cin, cout, dt)
    wk_n, bias_n = _prepare(wk, bias, nt, cin, 0, n, cin, cout)
    y1, s1, k1 = _in_fwd(a, None, 0, wk_n, bias_n, n, h, cin, cout, dt, nt_x=nt, norm_mode=ops.NORM_SCALED)
    assert k1.rstrip(">").endswith(", 2"), k1
    ok, err = _close(y1, y0, dt)
    assert ok, (k1, err)
    # the border rows / columns on their own (a wrong `ring` would hide in the rel-L2 of the whole tensor)
    for sl in (np.s_[:, 0], np.s_[:, -1], np.s_[:, :, 0], np.s_[:, :, -1]):
        ok, err = _close(y1[sl], y0[sl], dt)
        assert ok, (k1, sl, err)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant,n,h,cu,cs,cout", [("halo64_st", 2, 32, 64, 64, 64), ("halo128_st", 2, 16, 128, 128, 128), ("halo128_st", 1, 32, 256, 256, 256)])
def test_concat_scaled_mode(variant, n, h, cu, cs, cout, dt):
    ops = _ops()
    rng = np.random.default_rng(22)
    u = _t(rng.standard_normal((n, h, h, cu)), dt)
    a, stats, beta, nt, ahat = _block(rng, n, h, cs, dt)
    cin = cu + cs
    wk, bias = _wk(rng, cin, cout, dt), _t(rng.standard_normal(cout) * 0.1, "f32")
    ops.set_tuning("tapgemm.variant", variant)
    y0, _, _ = _in_fwd(u, ahat, cu, wk, bias, n, h, cin, cout, dt)
    wk_n, bias_n = _prepare(wk, bias, nt, cs, cu, n, cin, cout)
    # the un-folded source's weights are copied, the folded one's scaled
    assert torch.equal(wk_n[:, :, :, :cu], wk[None, :, :, :cu].expand(n, -1, -1, -1))
    y1, _, k1 = _in_fwd(u, a, cu, wk_n, bias_n, n, h, cin, cout, dt, nt_x2=nt, norm_mode=ops.NORM_SCALED)
    ok, err = _close(y1, y0, dt)
    assert ok, (k1, err)


@pytest.mark.parametrize("dt,rows", [("f32", 0), ("bf16", 4), ("bf16", 2), ("f32x3", 2)])       # f32x3: "wgrad.f32_split" (csrc/conv_wgrad_x3.hip)
@pytest.mark.parametrize("n,h,cin,cout,c1", [(3, 16, 64, 64, 0), (2, 32, 128, 64, 0), (5, 16, 64, 128, 0), (2, 48, 64, 64, 0), (2, 32, 192, 64, 64)])
def test_wgrad_scaled_mode(dt, rows, n, h, cin, cout, c1):
    """sum x_hat * dz = inv * sum a_ext * dz (kernel: `ring` outside the image, slab rows times inv, splits on sample boundaries)
    + (beta - mean * inv) * sum dz (shm_conv2d_wgrad_norm_finish from the per-sample dz sums)."""
    ops = _ops()
    rng = np.random.default_rng(23)
    cs = cin - c1
    u = _t(rng.standard_normal((n, h, h, c1)), dt) if c1 else None
    a, stats, beta, nt, ahat = _block(rng, n, h, cs, dt)
    dy = _t(rng.standard_normal((n, h, h, cout)) + 0.05, dt)
    if rows:
        ops.set_tuning("wgrad.bf16_rows", rows)
    key = "nt_x2" if c1 else "nt_x"
    src0, src1 = ((u, ahat), (u, a)) if c1 else ((ahat, None), (a, None))
    dw0, k0 = _wgrad(src0[0], src0[1], c1, dy, n, h, cin, cout)
    ws = torch.empty(ops.conv2d_wgrad_norm_workspace(n, h, h, cin, cout, 3, a.dtype) // 4 + 16, device="cuda")
    dw1 = torch.full((3, 3, cin, cout), 3.0, device="cuda")
    ops.conv2d_wgrad(src1[0], src1[1], c1, src1[0].shape[-1], 0 if src1[1] is None else src1[1].shape[-1], dy, cout, dw1, n, h, h, cin, cin, cout, 3, 1, 0,
                     ws, norm_mode=ops.NORM_SCALED, **{key: nt})
    k1 = ops.last_kernel()
    dzsum = dy.double().sum(dim=(1, 2)).contiguous()
    ops.conv2d_wgrad_norm_finish(dw1, nt, dzsum, n, cs, c1, cin, cout, 3)
    torch.cuda.synchronize()
    assert k1.rstrip(">").endswith("2"), k1
    g, r = dw1.double(), dw0.double()
    rel = float((g - r).norm() / r.norm())
    assert rel < (2e-5 if dt == "f32" else 1.5e-2), (k1, rel)
    if c1:                      # the un-folded source's rows are the plain kernel's sums (other split, same operands)
        assert float((g[:, :, :c1] - r[:, :, :c1]).norm() / r[:, :, :c1].norm()) < (2e-6 if dt == "f32" else 1e-6)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("gsum", [False, True])
def test_in_bwd_keeps_the_per_sample_dz_sums(dt, gsum):
    ops = _ops()
    rng = np.random.default_rng(24)
    n, h, c = 3, 32, 64
    a, stats, beta, nt, ahat = _block(rng, n, h, c, dt)
    g = _t(rng.standard_normal((n, h, h, c)), dt)
    dz = torch.empty_like(a)
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    keep = torch.full((n, c), 7.0, dtype=torch.float64, device="cuda")
    ops.in_bwd_keep_dz_sums(keep)
    if gsum:
        red = torch.zeros(ops.GSUM_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
        g64, a64 = g.double(), a.double()
        red.view(ops.GSUM_SLOTS, n, c, 2)[0] = torch.stack([g64.sum(dim=(1, 2)), (g64 * a64).sum(dim=(1, 2))], -1)
        dstage = torch.zeros(n * c, dtype=torch.float64, device="cuda")
        ops.in_bwd_apply(g, c, None, 0, a, c, stats, beta, red, None, dstage, dz, c, db, n, h, h, c, 0.2)
    else:
        red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
        ops.in_bwd(g, c, None, 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2)
    torch.cuda.synchronize()
    ref = dz.double().sum(dim=(1, 2))
    # (the sums are taken before dz is rounded to its storage type: fp32 partial sums in float32, plus bf16 rounding of the summands in bf16)
    assert float((keep - ref).abs().max()) < (1e-4 if dt == "f32" else 1e-2) * float(ref.abs().max() + 1.0)
    assert float((keep.sum(0) - db).abs().max()) < 1e-9 * float(db.abs().max() + 1.0)
    # one-shot: a second call leaves the buffer alone
    keep.fill_(7.0)
    db2 = torch.zeros_like(db)
    if not gsum:
        ops.in_bwd(g, c, None, 0, a, c, stats, red, dz, c, db2, n, h, h, c, 0.2)
        torch.cuda.synchronize()
        assert float(keep.min()) == 7.0


def _step(S, F, B, dt, fold, seed=0, mode=0):
    from shmgan_amd import ShmGANwithSSpecSeg
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype=dt).build()
    m.G.fold = fold
    m.G.norm_mode = mode
    m.G._plans.clear()
    inp = st.make_inputs(B, S)
    dr = st.make_draws(seed, B, S, F)
    m.train_step(*inp, draws=dr, style_factor=st.style_factor_intended(S), apply=False)
    torch.cuda.synchronize()
    plans = {k: dict(v) for k, v in m.G._plans.items()}
    return dict(m.losses()), m.G.P.grad.clone(), m.D.P.grad.clone(), plans


@pytest.mark.parametrize("S,F,B,dt", [(64, 64, 2, "float32"), (64, 64, 2, "bfloat16"), (128, 64, 1, "float32"), (64, 32, 2, "bfloat16")])
def test_train_step_is_bit_identical_with_and_without_the_fold(S, F, B, dt):
    # the folding instantiations of the bf16 weights-in-registers kernel are four-wave ones: compare against the four-wave plain kernel,
    # not against the eight-wave form the plain launches take by default ("tapgemm.wreg16": another accumulation order)
    _ops().set_tuning("tapgemm.wreg16", 0)
    l0, g0, d0, p0 = _step(S, F, B, dt, fold=False)
    l1, g1, d1, p1 = _step(S, F, B, dt, fold="all")
    assert not any(any(v.values()) for v in p0.values())
    folded = sorted({li for v in p1.values() for li, on in v.items() if on})
    assert folded, "no block was folded: the test would compare a path with itself"
    tol = 1e-6 if dt == "float32" else 2e-3      # bf16: a last-bit difference in a statistic can move a rounded activation
    for k, v in l0.items():
        if k != "ssim":
            assert abs(l1[k] - v) <= tol * max(1.0, abs(v)), (k, v, l1[k])
    assert rel_l2(host(g1), host(g0)) <= tol, (folded, rel_l2(host(g1), host(g0)))
    assert rel_l2(host(d1), host(d0)) <= tol


def test_default_policy_folds_the_float32_wreg_consumers():
    """"auto": at filter_size 64 the two blocks of the 256 x 256 level whose consumer is the fp32 weights-in-registers kernel are
    folded (encoder block 1 -> conv2d_1, the top Concatenate block -> conv2d_25), nothing in bfloat16."""
    from shmgan_amd import ShmGANwithSSpecSeg
    for dt, want in (("float32", [0, 20]), ("bfloat16", [])):
        m = ShmGANwithSSpecSeg(image_size=64, filter_size=64, batch_size=1, compute_dtype=dt).build()
        assert m.G.fold == "auto"
        plan = m.G._fold_plan(5, 5, False)
        assert sorted(li for li, on in plan.items() if on) == want, (dt, plan)


@pytest.mark.parametrize("S,F,B,dt", [(64, 64, 2, "float32"), (128, 64, 1, "float32"), (64, 64, 2, "bfloat16")])
def test_train_step_with_the_fold_in_the_operands(S, F, B, dt):
    """SHM_NORM_SCALED through the whole step (per-sample weights, bias rows, scaled slabs, rank-n term, per-sample dz sums): equal to the
    un-folded step to rounding.  Not to the run-to-run bound: a re-associated fp32 sum differs in the last bit, and at random
    initialisation the step amplifies a 1e-7 perturbation of an early activation about a thousandfold (LeakyReLU kinks, InstanceNorm
    of near-constant channels) -- the bound here is the oracle contract's (losses 1e-4, gradients 1e-3)."""
    l0, g0, d0, p0 = _step(S, F, B, dt, fold=False)
    l1, g1, d1, p1 = _step(S, F, B, dt, fold="all", mode=_ops().NORM_SCALED)
    folded = sorted({li for v in p1.values() for li, on in v.items() if on})
    assert folded
    # bf16: two paths that round at different points are each within the bf16 contract of the exact step (cosine >= 0.99), i.e.
    # within rel-L2 ~ 0.15 of each other
    ltol, gtol = (1e-4, 1e-3) if dt == "float32" else (2e-2, 0.15)
    for k, v in l0.items():
        if k != "ssim":
            assert abs(l1[k] - v) <= ltol * max(1.0, abs(v)), (k, v, l1[k])
    assert rel_l2(host(g1), host(g0)) <= gtol, (folded, rel_l2(host(g1), host(g0)))
    assert rel_l2(host(d1), host(d0)) <= gtol
    if dt != "float32":
        from util import cosine
        assert cosine(host(g1), host(g0)) >= 0.99


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("n,h,c,pool", [(3, 32, 64, False), (2, 48, 32, True), (2, 16, 256, False), (1, 24, 8, False)])
def test_in_bwd_pixel_mappings_agree(dt, n, h, c, pool):
    """"elem.interleave": a sample's blocks take pixel tiles round-robin (default) or one contiguous chunk each -- the same pixels either
    way (ragged tails included: 24 x 24 and 48 x 48 maps are not whole tiles), so dz and the bias gradient agree to the order of the sums."""
    ops = _ops()
    rng = np.random.default_rng(31)
    a, stats, beta, nt, ahat = _block(rng, n, h, c, dt)
    g = _t(rng.standard_normal((n, h, h, c)), dt)
    g2 = _t(rng.standard_normal((n, h // 2, h // 2, c)), dt) if pool else None
    outs = []
    for il in (0, 1):
        ops.set_tuning("elem.interleave", il)
        dz = torch.full_like(a, 9.0)
        db = torch.zeros(c, dtype=torch.float64, device="cuda")
        red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
        ops.in_bwd(g, c, g2, c if pool else 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2)
        torch.cuda.synchronize()
        assert float(red.abs().max()) == 0.0
        outs.append((dz.double(), db.clone()))
    (z0, b0), (z1, b1) = outs
    tol = 1e-5 if dt == "f32" else 2e-2
    assert float((z0 - z1).norm() / z0.norm()) < tol
    assert float((b0 - b1).abs().max()) < tol * float(b0.abs().max() + 1.0)
