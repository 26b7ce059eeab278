"""bf16 InstanceNorm backward in one pass (round 5: in_bwd_fused8_kernel, include/shmgan_hip.h: shm_in_bwd_fused_scratch; the backward of
SHM.py:244-245's Conv -> LeakyReLU -> InstanceNormalization block).

The one-pass form keeps a block's slice of the gradient and of the activation in registers between the reduce and the apply phase; the blocks
of a sample meet at a per-sample barrier.  Checked here:
  * against shm_in_bwd's two passes ("elem.fused_bwd" = 0) on the same operands: dz equal to bf16 rounding, bias gradient to the order of
    the float64 sums -- one source and the pooled form, 64 .. 512 channels, 1 .. 256 blocks per sample, many samples;
  * against a float64 restatement of the InstanceNorm + LeakyReLU backward;
  * the scratch is clean again on return (means, both counters and the flags of every sample, the timeout word), and there are no float
    atomics: repeated calls on one scratch give the same bits for dz and the bias gradient;
  * shapes the form does not take (ragged maps, more slices per group than twice fit the chip, fp32) run the two passes, and say so in shm_last_kernel;
  * round 6: in_bwd_fusedg_kernel (the gradient held in registers, the activation streamed twice) on the same contract and the same scratch.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops():
    from shmgan_amd import ops
    return ops


@pytest.fixture(autouse=True)
def _reset_tuning():
    yield
    _ops().set_tuning("reset", 0)


def _operands(rng, n, h, c, pool, gshift=0.0):
    """gshift: mean of the gradient.  With a mean of 25 on noise of 1 a sum that missed ONE block's contribution (a stale slot read behind the
    sample barrier) moves dz by ~10 % of its norm; the zero-mean operands would hide it."""
    ops = _ops()
    a = torch.from_numpy((rng.standard_normal((n, h, h, c)) * rng.uniform(0.5, 2.0, (n, 1, 1, c)) + rng.uniform(-1, 1, (n, 1, 1, c))).astype(np.float32)).cuda().to(BF)
    g = torch.from_numpy((rng.standard_normal((n, h, h, c)) + gshift).astype(np.float32)).cuda().to(BF)
    g2 = torch.from_numpy(rng.standard_normal((n, h // 2, h // 2, c)).astype(np.float32)).cuda().to(BF) if pool else None
    stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    return a, g, g2, stats


def _clean(scratch, n, h, c):
    """Zero on return: everything behind the per-block partial rows (means, both counters and the flags of every sample, the timeout word)."""
    rows = (n * (h * h * min(c, 64) // 16384) * 3 * c + 1) // 2
    tail = scratch[rows:]
    return torch.equal(tail.view(torch.int64), torch.zeros_like(tail).view(torch.int64))


def _run(a, g, g2, stats, n, h, c, fused, dbias=True):
    ops = _ops()
    dz = torch.full_like(a, 9.0)
    db = torch.zeros(c, dtype=torch.float64, device="cuda") if dbias else None
    red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    ops.in_bwd(g, c, g2, c if g2 is not None else 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=fused)
    kern = ops.last_kernel()
    torch.cuda.synchronize()
    assert float(red.abs().max()) == 0.0
    return dz, db, kern


def _reference(a, g, g2, stats, n, h, c, slope=0.2):
    """float64: xh = (a - mean) * inv; d = inv * (g - mean(g) - xh * mean(g * xh)); dz = d * lrelu'(a)."""
    A = a.double()
    G = g.double()
    if g2 is not None:
        G = G + 0.25 * g2.double().repeat_interleave(2, 1).repeat_interleave(2, 2)
    st = stats.view(n, c, 2)
    mean, inv = st[:, :, 0].view(n, 1, 1, c), st[:, :, 1].view(n, 1, 1, c)
    xh = (A - mean) * inv
    m1 = G.mean((1, 2), keepdim=True)
    m2 = (G * xh).mean((1, 2), keepdim=True)
    d = inv * (G - m1 - xh * m2)
    dz = torch.where(A > 0, d, d * slope)
    return dz, dz.sum((0, 1, 2))


# (n, h, c, pooled): barrier groups of 64 channels with 256-pixel slices (16 / 64 / 256 blocks per group; 128 .. 512 channels: 2 .. 8 groups per
# sample); 8 / 32 channels -> 2048 / 512-pixel slices; pooled: tiles of 2 x 128, 4 x 64, 16 x 16 pixels
SHAPES = [(3, 64, 64, False), (2, 128, 64, True), (2, 256, 64, False), (2, 256, 64, True), (2, 128, 128, True), (5, 64, 128, True), (3, 32, 256, False),
          (2, 16, 512, True), (2, 64, 32, False), (1, 64, 8, False), (41, 16, 64, False)]


@pytest.mark.parametrize("gshift", [0.0, 25.0])
@pytest.mark.parametrize("n,h,c,pool", SHAPES)
def test_fused_matches_two_passes_and_float64(n, h, c, pool, gshift):
    ops = _ops()
    rng = np.random.default_rng(7 + n + h + c)
    a, g, g2, stats = _operands(rng, n, h, c, pool, gshift)
    scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    ops.set_tuning("elem.fused_bwd", 0)
    z0, b0, k0 = _run(a, g, g2, stats, n, h, c, scratch)
    assert "fused" not in k0, k0
    ops.set_tuning("elem.fused_bwd", 1)
    z1, b1, k1 = _run(a, g, g2, stats, n, h, c, scratch)
    # automatic dispatch: the g-held kernel (round 6) from 256 eight-slot slices per group on, the kernel that holds g and a below
    big = not pool and h * h * min(c, 64) // 16384 >= 256
    assert k1 == ("in_bwd_fused8_kernel<true>" if pool else "in_bwd_fusedg_kernel<2, 2, 4>" if big else "in_bwd_fused8_kernel<false>"), k1
    assert _clean(scratch, n, h, c)
    # the two forms round the same float32 value to bf16 unless the last bits of the two means differ: a handful of one-ulp differences
    diff = (z0.float() - z1.float()).abs()
    assert float((diff > 0).float().mean()) < 0.02, float((diff > 0).float().mean())
    assert float(diff.double().norm() / z0.double().norm()) < 2e-3
    assert float((b0 - b1).abs().max()) <= 2e-3 * float(b0.abs().max() + 1.0)       # sums of bf16-rounded dz: a flipped ulp moves them
    zr, br = _reference(a, g, g2, stats, n, h, c)
    assert float((z1.double() - zr).norm() / zr.norm()) < 4e-3                          # bf16 output rounding: 2^-9 relative per element
    assert float((b1 - br).abs().max()) <= 4e-3 * float(zr.abs().sum((0, 1, 2)).max())
    # second call on the same scratch: no float atomics, the rows are added in block order -> the same bits
    z2, b2, _ = _run(a, g, g2, stats, n, h, c, scratch)
    assert torch.equal(z2, z1) and torch.equal(b2, b1)
    assert _clean(scratch, n, h, c)


def test_fused_without_bias_gradient():
    ops = _ops()
    rng = np.random.default_rng(3)
    n, h, c = 2, 64, 64
    a, g, g2, stats = _operands(rng, n, h, c, False)
    scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    z1, _, k1 = _run(a, g, None, stats, n, h, c, scratch, dbias=False)
    assert k1 == "in_bwd_fused8_kernel<false>"
    zr, _ = _reference(a, g, None, stats, n, h, c)
    assert float((z1.double() - zr).norm() / zr.norm()) < 4e-3
    assert _clean(scratch, n, h, c)


# Round 6: in_bwd_fusedg_kernel -- g held in registers, a streamed twice, 16 pixel slots per thread.  Forced ("elem.fused_hold" = 2) on every shape that
# has whole 32768 / CB-pixel slices, both register-budget forms, against the kernel that holds both tensors (same scratch: ONE layout), the two passes
# and float64; the 512 x 512 x 64 map of BASELINE configs[3] (512 blocks per group) is taken automatically.
GSHAPES = [(3, 64, 64), (2, 256, 64), (1, 512, 64), (2, 128, 128), (3, 32, 256), (2, 32, 512), (2, 64, 32), (1, 64, 8), (41, 32, 64)]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("gshift", [0.0, 25.0])
@pytest.mark.parametrize("n,h,c", GSHAPES)
def test_g_held_kernel_matches_the_other_forms(n, h, c, gshift, variant):
    ops = _ops()
    rng = np.random.default_rng(17 + n + h + c)
    a, g, _, stats = _operands(rng, n, h, c, False, gshift)
    scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    ops.set_tuning("elem.fused_gvariant", variant)
    name = "in_bwd_fusedg_kernel<8, 8, 3>" if variant else "in_bwd_fusedg_kernel<2, 2, 4>"
    groups_fit = 2 * (h * h * min(c, 64) // 32768) <= (768 if variant else 1024)          # twice a group's blocks on a whole MI355X
    ops.set_tuning("elem.fused_hold", 2)
    zg, bg, kg = _run(a, g, None, stats, n, h, c, scratch)
    if not groups_fit:
        assert "fusedg" not in kg, kg
        return
    assert kg == name, kg
    assert _clean(scratch, n, h, c)
    ops.set_tuning("elem.fused_hold", 1)                       # the round-5 kernel on the SAME scratch (where it takes the shape)
    z8, b8, k8 = _run(a, g, None, stats, n, h, c, scratch)
    assert _clean(scratch, n, h, c)
    ops.set_tuning("elem.fused_bwd", 0)
    z0, b0, k0 = _run(a, g, None, stats, n, h, c, scratch)
    assert "fused" not in k0
    for zo, bo in ((z8, b8), (z0, b0)):
        diff = (zo.float() - zg.float()).abs()
        assert float((diff > 0).float().mean()) < 0.02
        assert float(diff.double().norm() / zo.double().norm()) < 2e-3
        assert float((bo - bg).abs().max()) <= 2e-3 * float(bo.abs().max() + 1.0)
    zr, br = _reference(a, g, None, stats, n, h, c)
    assert float((zg.double() - zr).norm() / zr.norm()) < 4e-3
    assert float((bg - br).abs().max()) <= 4e-3 * float(zr.abs().sum((0, 1, 2)).max())
    ops.set_tuning("elem.fused_bwd", 1)
    ops.set_tuning("elem.fused_hold", 2)
    z2, b2, _ = _run(a, g, None, stats, n, h, c, scratch)        # bitwise repeatable, also right behind the other kernels on this scratch
    assert torch.equal(z2, zg) and torch.equal(b2, bg)
    assert _clean(scratch, n, h, c)


def test_g_held_kernel_is_the_automatic_choice_on_the_largest_maps():
    ops = _ops()
    rng = np.random.default_rng(23)
    for n, h, c, want in ((1, 512, 64, "in_bwd_fusedg_kernel<2, 2, 4>"), (2, 256, 64, "in_bwd_fusedg_kernel<2, 2, 4>"), (2, 256, 128, "in_bwd_fusedg_kernel<2, 2, 4>"),
                          (2, 128, 128, "in_bwd_fused8_kernel<false>"), (2, 64, 256, "in_bwd_fused8_kernel<false>")):
        a, g, _, stats = _operands(rng, n, h, c, False)
        scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
        z, b, k = _run(a, g, None, stats, n, h, c, scratch)
        assert k == want, (n, h, c, k)
        zr, _ = _reference(a, g, None, stats, n, h, c)
        assert float((z.double() - zr).norm() / zr.norm()) < 4e-3 and _clean(scratch, n, h, c)


@pytest.mark.parametrize("n,h,c,why", [(2, 24, 64, "ragged: 576 pixels are not whole 256-pixel slices"),
                                       (1, 1024, 64, "4096 / 2048 slices per group: more than a resident group"),
                                       (1, 64, 8, "pooled form: 64-channel groups only"),
                                       (2, 8, 64, "a 64-pixel map is smaller than one slice"),
                                       (2, 32, 24, "3 channel lanes do not divide a block")])
def test_shapes_outside_the_form_run_two_passes(n, h, c, why):
    ops = _ops()
    rng = np.random.default_rng(11)
    pool = why.startswith("pooled")
    a, g, g2, stats = _operands(rng, n, h, c, pool)
    scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    z1, b1, k1 = _run(a, g, g2, stats, n, h, c, scratch)
    assert "fused" not in k1, (k1, why)
    zr, br = _reference(a, g, g2, stats, n, h, c)
    assert float((z1.double() - zr).norm() / zr.norm()) < 4e-3
    assert float(scratch.abs().max()) == 0.0


def test_small_scratch_and_fp32_run_two_passes():
    ops = _ops()
    rng = np.random.default_rng(12)
    n, h, c = 2, 64, 64
    a, g, g2, stats = _operands(rng, n, h, c, False)
    small = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c) - 1, dtype=torch.float64, device="cuda")
    _, _, k = _run(a, g, None, stats, n, h, c, small)
    assert "fused" not in k
    af, gf = a.float(), g.float()
    scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    _, _, k = _run(af, gf, None, stats, n, h, c, scratch)
    assert "fused" not in k
    # the request is one-shot: a call without `fused` right after one with it runs two passes
    _, _, k = _run(a, g, None, stats, n, h, c, scratch)
    assert "fused" in k
    _, _, k = _run(a, g, None, stats, n, h, c, None)
    assert "fused" not in k
