"""The fused block's backward (round 3): InstanceNorm-backward sums taken in the epilogue of the launch that WRITES the
gradient ("gsum", include/shmgan_hip.h: shm_conv2d_dgrad_gsum / shm_conv2d_fwd_gsum / shm_in_bwd_apply).

Every kernel family with a gsum epilogue is forced through shm_set_tuning and checked three ways:
  * the gradient tensor it writes is BIT-IDENTICAL to the plain entry point's (the sums are a side output);
  * the slot sums equal (sum g, sum g * aux) of the stored values, recomputed in float64 on the host;
  * shm_in_bwd_apply fed with those sums reproduces shm_in_bwd (reduce + apply passes) to rounding, also for the pooled form
    (gradient of AveragePooling2D summed against the pooled normalised tensor), and leaves every f64 scratch zero.
Variants without a gsum epilogue (the fused four-phase kernel, tiny maps) must deliver the same sums through the follow-up
reduce pass.  SHM.py:244-245 (Conv -> LeakyReLU -> InstanceNorm block order) is what makes d_out the output of the next
layer's input-gradient product.
"""
import numpy as np
import pytest
import torch

from util import host, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops():
    from shmgan_amd import ops
    return ops


@pytest.fixture(autouse=True)
def _reset_tuning():
    yield
    _ops().set_tuning("reset", 0)


def _t(a, dt):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    return t.to(BF) if dt == "bf16" else t


def _sums(g, aux):
    """float64 (sum g, sum g * aux) per (sample, channel) of the tensors as the device holds them."""
    g64, a64 = g.double().cpu().numpy(), aux.double().cpu().numpy()
    n, c = g64.shape[0], g64.shape[-1]
    g2, a2 = g64.reshape(n, -1, c), a64.reshape(n, -1, c)
    return np.stack([g2.sum(1), (g2 * a2).sum(1)], -1)          # [n, c, 2]


def _red_total(red, n, c):
    return host(red).reshape(_ops().GSUM_SLOTS, n, c, 2).sum(0)


def _check_sums(red, g, aux, dt):
    n, c = g.shape[0], g.shape[-1]
    got, ref = _red_total(red, n, c), _sums(g, aux)
    # float32 partial sums inside a wave tile, float64 across tiles: relative to the scale of the summands
    scale = np.sqrt((g.double().cpu().numpy() ** 2).reshape(n, -1, c).sum(1))[..., None] + 1e-30
    assert (np.abs(got - ref) / scale).max() < 2e-5, (np.abs(got - ref) / scale).max()


def _dgrad_case(variant, dt, n, h, cin, cout, n1, stride=1, k=3, which=(True, True)):
    """conv2d_dgrad under a forced variant: cout -> cin input channels (the GEMM's N), optionally split (n1) into two output
    tensors with their own aux / sums."""
    ops = _ops()
    rng = np.random.default_rng(7)
    ho = h // stride
    adt = BF if dt == "bf16" else torch.float32
    dy = _t(rng.standard_normal((n, ho, ho, cout)), dt)
    w = _t(rng.standard_normal((k, k, cin, cout)) * 0.1, dt)
    split = 0 < n1 < cin
    c0, c1 = (n1, cin - n1) if split else (cin, 0)
    aux0 = _t(rng.standard_normal((n, h, h, c0)), dt)
    aux1 = _t(rng.standard_normal((n, h, h, c1)), dt) if split else None
    outs = []
    for use_gsum in (False, True):
        dx = torch.full((n, h, h, c0), 7.0, device="cuda", dtype=adt)
        dx2 = torch.full((n, h, h, c1), 7.0, device="cuda", dtype=adt) if split else None
        red0 = torch.zeros(ops.GSUM_SLOTS * n * c0 * 2, dtype=torch.float64, device="cuda")
        red1 = torch.zeros(ops.GSUM_SLOTS * n * c1 * 2, dtype=torch.float64, device="cuda") if split else None
        ops.set_tuning("tapgemm.variant", variant)
        # the bit-for-bit comparison below is between the gsum form and the plain form of the SAME kernel: in bf16 the plain
        # weights-in-registers launch defaults to tapgemm_wreg16_bf16_kernel (round 4), which walks halo rows -- another fp32 summation
        # order than the four-wave kernel the gsum form lives in (equal to rounding, not to the bit); pin the plain call to that kernel
        if variant == "wreg" and dt == "bf16":
            ops.set_tuning("tapgemm.wreg16", 0)
        g0 = (aux0, c0, red0) if (use_gsum and which[0]) else None
        g1 = (aux1, c1, red1) if (use_gsum and split and which[1]) else None
        if use_gsum and g0 is None and g1 is None:
            g0 = (aux0, c0, red0)
        ops.conv2d_dgrad(dy, cout, w, dx, dx2, n1 if split else 0, c0, c1, n, h, h, cin, cout, k, stride, gsum=g0, gsum2=g1)
        torch.cuda.synchronize()
        outs.append((dx, dx2, red0, red1, g0, g1, ops.last_kernel()))
    (dxa, dx2a, *_), (dxb, dx2b, red0, red1, g0, g1, kern) = outs
    assert torch.equal(dxa, dxb) and (not split or torch.equal(dx2a, dx2b)), kern           # the sums do not touch the gradient
    if g0 is not None:
        _check_sums(red0, dxb, aux0, dt)
    if g1 is not None:
        _check_sums(red1, dx2b, aux1, dt)
    return kern


HALO = ["halo128", "halo64", "halo128_st", "halo64_st"]
DMA = ["dma128x128", "dma64x128", "dma128x64", "dma64x64", "dma256x64", "dma256x128", "dma128x128_bk32", "dma128x128_nst4"]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant", HALO + DMA)
@pytest.mark.parametrize("n,h,cin,cout,n1,which", [
    (3, 16, 64, 64, 0, (True, True)),            # one part
    (2, 32, 256, 64, 128, (False, True)),        # Concatenate's gradient: sums only for the skip half (the decoder's case)
    (2, 16, 192, 128, 64, (True, True)),         # both parts, ragged N for the 128-wide tiles
])
def test_dgrad_gsum_forced_variant(variant, dt, n, h, cin, cout, n1, which):
    _dgrad_case(variant, dt, n, h, cin, cout, n1, which=which)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant", ["dma128x128", "dma64x128", "dma256x128", "halo128_st"])
def test_dgrad_gsum_with_outputs_beyond_4gib(variant, dt):
    """"tapgemm.flat_epilogue": the DMA tiles keep their sums on the 64-bit-address epilogue, the halo kernels (32-bit aux offsets) hand
    them to the follow-up reduce pass -- same gradient, same sums either way"""
    _ops().set_tuning("tapgemm.flat_epilogue", 1)
    _dgrad_case(variant, dt, 2, 32, 256, 64, 128, which=(True, True))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("n,h,cin,cout,n1,which", [
    (3, 16, 64, 64, 0, (True, True)),            # 64 <- 64: the layers in front of the 256 x 256 blocks
    (2, 32, 128, 64, 64, (False, True)),         # 128 <- 64 with the skip half summed
    (7, 48, 64, 64, 0, (True, True)),            # blocks that cross image boundaries (carried sums are flushed per image)
    (2, 32, 128, 32, 64, (True, True)),          # K = 32 (bf16) / two 16-channel chunks (fp32)
])
def test_dgrad_gsum_wreg(dt, n, h, cin, cout, n1, which):
    kern = _dgrad_case("wreg", dt, n, h, cin, cout, n1, which=which)
    assert kern.startswith("tapgemm_wreg") and kern.endswith("true>"), kern          # the gsum instantiation ran


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_dgrad_gsum_stride2_and_fallback(dt):
    """Stride-2 input gradients: the fused four-phase kernel has no gsum epilogue (follow-up reduce pass), the DMA tiles take
    the sums per phase; a 4 x 4 map (16 pixels per sample) has no whole wave tile per sample: reduce pass again."""
    assert _dgrad_case("phase4", dt, 2, 32, 64, 128, 0, stride=2).startswith("tapgemm_phase4")
    _dgrad_case("dma128x128", dt, 2, 32, 64, 128, 0, stride=2)
    _dgrad_case("dma128x64", dt, 3, 16, 64, 64, 0, stride=2)
    _dgrad_case("auto", dt, 3, 4, 64, 64, 0)
    _dgrad_case("auto", dt, 2, 8, 128, 128, 0, k=1)                                  # the 1 x 1 bottleneck


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("variant", ["auto", "dma128x128", "dma256x128", "dma64x64"])
def test_fwd_gsum_stride2(variant, dt):
    """shm_conv2d_fwd_gsum on the stride-2 forward form (= Conv2DTranspose's input gradient in model.py)."""
    ops = _ops()
    rng = np.random.default_rng(9)
    n, h, cin, cout = 2, 32, 64, 128
    adt = BF if dt == "bf16" else torch.float32
    x = _t(rng.standard_normal((n, h, h, cin)), dt)
    w = rng.standard_normal((3, 3, cin, cout)) * 0.1
    wk = torch.zeros(9 * cout * cin, device="cuda", dtype=adt)
    ops.transpose_taps(torch.from_numpy(w.astype(np.float32)).cuda(), wk, 9, cin, cout, cin)
    aux = _t(rng.standard_normal((n, h // 2, h // 2, cout)), dt)
    ya = torch.empty((n, h // 2, h // 2, cout), device="cuda", dtype=adt)
    yb = torch.empty_like(ya)
    red = torch.zeros(ops.GSUM_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
    ops.set_tuning("tapgemm.variant", variant)
    ops.conv2d_fwd(x, None, 0, cin, 0, wk, None, ya, cout, n, h, h, cin, cout, 3, 2, 1.0)
    ops.conv2d_fwd(x, None, 0, cin, 0, wk, None, yb, cout, n, h, h, cin, cout, 3, 2, 1.0, gsum=(aux, cout, red))
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)
    _check_sums(red, yb, aux, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("pooled", [False, True])
@pytest.mark.parametrize("n,h,c", [(3, 16, 64), (2, 32, 128), (5, 8, 256)])
def test_in_bwd_apply_matches_reduce_plus_apply(dt, pooled, n, h, c):
    """One pass with the producer's sums == shm_in_bwd's two passes.  The sums are built here by the stand-alone reduce path
    (a dgrad of a 4 x 4-map layer cannot be arranged for every shape; the forced-variant tests above tie the epilogues to the
    same sums), in the raw form against a and in the pooled form against avgpool(normalised a)."""
    ops = _ops()
    rng = np.random.default_rng(11)
    adt = BF if dt == "bf16" else torch.float32
    a = _t(rng.standard_normal((n, h, h, c)) * 1.5 + 0.3, dt)
    g1 = _t(rng.standard_normal((n, h, h, c)), dt)
    g2 = _t(rng.standard_normal((n, h // 2, h // 2, c)), dt) if pooled else None
    beta = torch.from_numpy(rng.normal(0, 0.02, c).astype(np.float32)).cuda()
    stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    ahat = torch.empty_like(a)
    pool = torch.empty((n, h // 2, h // 2, c), device="cuda", dtype=adt)
    ops.in_apply_pool(a, c, stats, beta, ahat, c, pool, c, n, h, h, c)
    # reference: reduce + apply
    red3 = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    dz_ref = torch.empty_like(a)
    db_ref = torch.zeros(c, dtype=torch.float64, device="cuda")
    ops.in_bwd(g1, c, g2, c, a, c, stats, red3, dz_ref, c, db_ref, n, h, h, c, 0.2)
    # sums as the producers deliver them: exact float64 sums of the stored tensors, placed in slot 0 of the slot copies
    red = torch.zeros(ops.GSUM_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
    redp = torch.zeros_like(red) if pooled else None

    def reduce_into(g, aux, hw, dst):
        dst.view(ops.GSUM_SLOTS, n, c, 2)[0].copy_(torch.from_numpy(_sums(g, aux)).cuda())

    reduce_into(g1, a, h * h, red)
    if pooled:
        reduce_into(g2, pool, h * h // 4, redp)
    dstage = torch.zeros(n * c, dtype=torch.float64, device="cuda")
    dz = torch.empty_like(a)
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    ops.in_bwd_apply(g1, c, g2, c, a, c, stats, beta, red, redp, dstage, dz, c, db, n, h, h, c, 0.2)
    torch.cuda.synchronize()
    tol = 2e-2 if dt == "bf16" and pooled else (4e-3 if dt == "bf16" else 2e-5)
    assert rel_l2(host(dz.float()), host(dz_ref.float())) < tol, rel_l2(host(dz.float()), host(dz_ref.float()))
    # bias gradient = sum of dz (heavy cancellation): compared on the scale of the summands
    scale = host(dz_ref.float().abs()).reshape(-1, c).sum(0)
    assert (np.abs(host(db) - host(db_ref)) / scale).max() < (4e-3 if dt == "bf16" else 2e-6)
    assert float(red.abs().max()) == 0.0 and float(dstage.abs().max()) == 0.0 and (redp is None or float(redp.abs().max()) == 0.0)


@pytest.mark.parametrize("dt", ["float32", "bfloat16"])
@pytest.mark.parametrize("S,F,B", [(64, 32, 2), (64, 64, 1)])
def test_whole_step_with_and_without_gsum(dt, S, F, B):
    """The same train_step with the InstanceNorm-backward sums taken in the epilogues (model.gsum = True: the float32 default)
    and by shm_in_bwd's reduce pass (False: the bfloat16 default): identical forward, gradients equal to rounding.  Keeps the
    non-default combination of either dtype under test (the parity suites run each dtype with its default)."""
    from oracle import step_torch as st
    from shmgan_amd import ShmGANwithSSpecSeg
    res = {}
    for on in (False, True):
        m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype=dt).build()
        m.G.gsum = m.D.gsum = on
        m.train_step(*st.make_inputs(B, S), draws=st.make_draws(3, B, S, F), style_factor=st.style_factor_intended(S), apply=False)
        torch.cuda.synchronize()
        res[on] = (dict(m.losses()), host(m.G.P.grad), host(m.D.P.grad))
    for k, v in res[False][0].items():                            # the forward pass does not depend on it (run-to-run bound: f64 atomics)
        if k != "ssim":
            assert abs(res[True][0][k] - v) <= 1e-6 * max(1.0, abs(v)), (k, v, res[True][0][k])
    tol = 2e-5 if dt == "float32" else 2e-2
    for i in (1, 2):
        assert rel_l2(res[True][i], res[False][i]) < tol, (i, rel_l2(res[True][i], res[False][i]))
