"""Host-side logic of the consumer-side InstanceNorm fold that needs no GPU: the eligibility queries are dry runs of the launchers'
variant choice (include/shmgan_hip.h: shm_conv2d_norm_supported, shm_conv2d_wgrad_norm_supported), the scaled mode's workspace query
and the argument checks of the new entry points.  (What the kernels compute is tests/test_norm_fold_gpu.py.)"""
import ctypes

import pytest

from shmgan_amd import _lib

F32, BF16 = 0, 1


@pytest.fixture(autouse=True)
def _reset():
    yield
    _lib.lib().shm_set_tuning(b"reset", 0)


def fwd_ok(*a):
    return bool(_lib.lib().shm_conv2d_norm_supported(*a))


def wg_ok(*a):
    return bool(_lib.lib().shm_conv2d_wgrad_norm_supported(*a))


@pytest.mark.parametrize("dt", [F32, BF16])
def test_forward_query_follows_the_variant_choice(dt):
    # (batch, hi, wi, cin, c1, cout, ksize, stride, norm_part, dtype)
    assert fwd_ok(40, 256, 256, 64, 0, 64, 3, 1, 0, dt)                 # weights-in-registers kernel
    assert fwd_ok(40, 128, 128, 128, 0, 128, 3, 1, 0, dt)               # 128-wide halo block
    assert fwd_ok(40, 256, 256, 128, 64, 64, 3, 1, 1, dt)               # Concatenate: the skip is the folded source
    assert fwd_ok(40, 64, 64, 512, 256, 256, 3, 1, 1, dt)               # 256 folded channels: the LDS table's limit
    assert not fwd_ok(40, 32, 32, 1024, 512, 512, 3, 1, 1, dt)          # 512 folded channels
    assert not fwd_ok(40, 256, 256, 64, 0, 128, 3, 2, 0, dt)            # stride 2: DMA tiles
    assert not fwd_ok(40, 16, 16, 512, 0, 512, 1, 1, 0, dt)             # 1x1
    assert not fwd_ok(40, 24, 24, 64, 0, 64, 3, 1, 0, dt)               # map not a multiple of 16
    assert not fwd_ok(40, 128, 128, 128, 0, 128, 3, 1, 1, dt)           # part 1 of a one-source convolution
    assert not fwd_ok(40, 128, 128, 128, 0, 128, 3, 1, 0, 7)            # unknown dtype


def test_forward_query_depends_on_the_grid_and_on_forced_variants():
    L = _lib.lib()
    # fp32, fewer 64-wide halo blocks than CUs: the launcher takes the 64 x 64 DMA tile, which cannot fold; bf16 keeps the halo block
    assert not fwd_ok(2, 32, 32, 128, 0, 128, 3, 1, 0, F32)
    assert fwd_ok(2, 32, 32, 128, 0, 128, 3, 1, 0, BF16)
    L.shm_set_tuning(b"tapgemm.variant", 12)          # SHM_TG_HALO128_ST forced
    assert fwd_ok(2, 32, 32, 128, 0, 128, 3, 1, 0, F32)
    L.shm_set_tuning(b"tapgemm.variant", 3)           # a DMA tile forced
    assert not fwd_ok(40, 128, 128, 128, 0, 128, 3, 1, 0, F32)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_wgrad_query(dt):
    # (batch, hi, wi, cin, cin_ld, c1, cout, ksize, stride, norm_part, dtype)
    assert wg_ok(40, 256, 256, 64, 64, 0, 64, 3, 1, 0, dt)
    assert wg_ok(40, 128, 128, 256, 256, 128, 128, 3, 1, 1, dt)
    assert not wg_ok(40, 128, 128, 192, 192, 96, 64, 3, 1, 1, dt)       # a 64-channel tile would straddle the two sources
    assert not wg_ok(40, 256, 256, 64, 64, 0, 128, 3, 2, 0, dt)         # stride 2: the generic kernel
    assert not wg_ok(40, 40, 40, 64, 64, 0, 64, 3, 1, 0, dt)            # map width not a multiple of 16
    L = _lib.lib()
    L.shm_set_tuning(b"wgrad.variant", 1)             # generic kernels only
    assert not wg_ok(40, 256, 256, 64, 64, 0, 64, 3, 1, 0, dt)


def test_thin_input_layer_takes_the_packed_kernel_not_the_fold():
    # fp32, 10 real channels in a 16-channel pitch: wgrad_halo_thin_kernel packs (tap, ci) into the MFMA rows and has no norm form
    assert not wg_ok(8, 64, 64, 10, 16, 0, 64, 3, 1, 0, F32)


@pytest.mark.parametrize("dt", [F32, BF16])
def test_scaled_mode_workspace_covers_sample_aligned_splits(dt):
    L = _lib.lib()
    for batch, h, cin, cout in ((40, 256, 64, 64), (8, 256, 64, 64), (40, 128, 128, 128), (5, 48, 64, 64), (3, 16, 64, 128)):
        plain = L.shm_conv2d_wgrad_workspace(batch, h, h, cin, cout, 3)
        aligned = L.shm_conv2d_wgrad_norm_workspace(batch, h, h, cin, cout, 3, dt)
        slab = 9 * cin * cout * 4
        assert aligned >= plain and aligned % slab == 0
        # never more slabs than patches, never more than twice the automatic split
        rows = 2 if dt == F32 else (4 if h % 4 == 0 else 2)
        assert aligned // slab <= batch * (h // rows) * (h // 16)
        assert aligned <= 2 * plain + slab * batch


def test_new_entry_points_check_their_arguments():
    L = _lib.lib()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    # both sources folded at once
    rc = L.shm_conv2d_in_fwd_norm(p, p, 64, 64, 64, p, p, 0, p, p, p, 64, 1, 16, 16, 128, 64, 3, 1, 0.2, p, None, 1e-6, None, None, F32, None)
    assert rc == -1 and b"at most one source" in L.shm_last_error()
    # a table for a second source that does not exist
    rc = L.shm_conv2d_in_fwd_norm(p, None, 0, 64, 0, None, p, 0, p, p, p, 64, 1, 16, 16, 64, 64, 3, 1, 0.2, p, None, 1e-6, None, None, F32, None)
    assert rc == -1 and b"nt_x2 without a second source" in L.shm_last_error()
    # unknown mode
    rc = L.shm_conv2d_in_fwd_norm(p, None, 0, 64, 0, p, None, 5, p, p, p, 64, 1, 16, 16, 64, 64, 3, 1, 0.2, p, None, 1e-6, None, None, F32, None)
    assert rc == -1 and b"norm_mode" in L.shm_last_error()
    # this block's own table without its beta
    rc = L.shm_conv2d_in_fwd_norm(p, None, 0, 64, 0, None, None, 0, p, p, p, 64, 1, 16, 16, 64, 64, 3, 1, 0.2, p, None, 1e-6, p, None, F32, None)
    assert rc == -1 and b"beta_out" in L.shm_last_error()
    rc = L.shm_conv2d_norm_prepare(p, None, p, 64, 96, p, p, 1, 128, 64, 3, F32, None)
    assert rc == -1 and b"outside" in L.shm_last_error()
    rc = L.shm_conv2d_wgrad_norm_finish(p, p, p, 1, 64, 96, 128, 64, 3, None)
    assert rc == -1 and b"outside" in L.shm_last_error()
    rc = L.shm_in_norm_table(None, p, p, 1, 64, None)
    assert rc == -1 and b"null pointer" in L.shm_last_error()
