"""Thin tensor-level wrappers over the C ABI (include/shmgan_hip.h).

PyTorch tensors are storage only: every function takes float32 CUDA(ROCm) tensors, passes
their device pointers to libshmgan_hip.so on torch's current stream and returns nothing
(outputs are preallocated by the caller).  No torch.nn op runs here.
"""
from __future__ import annotations

import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event timing of the MFMA conv launches on the stream they run on (bench.py
    `roofline`).  Enabled by setting ops.TIMER = KernelTimer(); records (kernel symbol,
    algorithmic flops, start event, end event) per launch.  The symbol is the one the entry point reports
    through shm_last_kernel() (plus a suffix when the timed region holds more than that kernel)."""

    def __init__(self):
        self.recs = []
        self.brecs = []          # HBM-bound passes: (entry point, algorithmic bytes, start event, end event)

    def wrap(self, sym, flops, fn, label="", fixed=False):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        if not fixed:
            k = lib().shm_last_kernel()               # the variant the entry point actually dispatched to
            sym = (k.decode() if k else "?") + sym
        self.recs.append((sym, flops, e0, e1, label))

    def wrap_bytes(self, name, nbytes, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        if callable(nbytes):                 # bytes that depend on the path the entry point took (known after the call)
            nbytes = nbytes()
        self.brecs.append((name, nbytes, e0, e1))

    def bytes_summary(self):
        """{entry point: dict(launches, ms, bytes)} of the HBM-bound passes (algorithmic bytes: every tensor the pass must read or
        write, once) -- call after a device synchronize."""
        out = {}
        for name, nb, e0, e1 in self.brecs:
            d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nb
        return out

    def per_shape(self):
        """{(symbol, label): dict(launches, ms, flops)} for tuning."""
        out = {}
        for sym, flops, e0, e1, label in self.recs:
            d = out.setdefault((sym, label), dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
        return out

    def summary(self):
        """{symbol: dict(launches, ms, flops)} -- call after a device synchronize."""
        out = {}
        for sym, flops, e0, e1, _ in self.recs:
            d = out.setdefault(sym, dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
        return out


TIMER = None


def _timed(sym, flops, fn, label=""):
    if TIMER is None:
        fn()
    else:
        TIMER.wrap(sym, flops, fn, label)


def _timed_bytes(name, nbytes, fn):
    if TIMER is None:
        fn()
    else:
        TIMER.wrap_bytes(name, nbytes, fn)


def _p(t):
    return 0 if t is None else t.data_ptr()


def _dt(t):
    """SHM_F32 / SHM_BF16 code of an activation tensor."""
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float32:
        return 0
    raise TypeError(f"activation tensors are float32 or bfloat16, got {t.dtype}")


def _dtg(act, grad):
    """dtype code of a call with activation tensor `act` and gradient-signal ([G]) tensor `grad`:
    SHM_BF16_GF32 when the activations are bf16 and the gradient signal is kept in fp32."""
    d = _dt(act)
    if d == 1 and grad is not None and grad.dtype == torch.float32:
        return 2
    if grad is not None and d != 2 and _dt(grad) != d:
        raise TypeError("gradient-signal tensors are float32, or share the activation dtype")
    return d


# SHM_TG_* of include/shmgan_hip.h
TAPGEMM_VARIANTS = {"auto": 0, "halo128": 1, "halo64": 2, "dma128x128": 3, "dma64x128": 4, "dma128x64": 5, "dma256x64": 6,
                    "dma256x128": 7, "halo128_ph8": 8, "dma128x128_bk32": 9, "dma128x128_nst4": 10, "wreg": 11, "halo128_st": 12, "halo64_st": 13, "phase4": 14, "dma64x64": 15, "halo128_st_w4": 16}


def set_tuning(key, value):
    """shm_set_tuning: dispatch knobs of the MFMA kernels ("tapgemm.variant", "wgrad.variant", ... see the header);
    `value` may be a TAPGEMM_VARIANTS name.  set_tuning("reset", 0) restores every default."""
    if isinstance(value, str):
        value = TAPGEMM_VARIANTS[value]
    check(lib().shm_set_tuning(key.encode(), int(value)), "shm_set_tuning")


def get_tuning(key):
    import ctypes
    v = ctypes.c_int(0)
    check(lib().shm_get_tuning(key.encode(), ctypes.addressof(v)), "shm_get_tuning")
    return v.value


def last_kernel():
    k = lib().shm_last_kernel()
    return k.decode() if k else ""


def cast_f32(src, dst, n):
    check(lib().shm_cast_f32(_p(src), _p(dst), n, _dt(dst), _stream()), "shm_cast_f32")


class TransposeBatch:
    """Host tables of shm_transpose_taps_multi for a fixed set of (w, wt, ntaps, rows, cols, rows_pad): built once (the
    tensors are persistent views), launched after every optimizer step."""

    def __init__(self, items):
        import ctypes as C
        self.items = list(items)              # keeps the tensors alive
        n = len(self.items)
        assert n <= 48
        self.n = n
        self.dt = _dt(self.items[0][1]) if n else 0
        assert all(_dt(it[1]) == self.dt for it in self.items)
        self.w = (C.c_void_p * n)(*[it[0].data_ptr() for it in self.items])
        self.wt = (C.c_void_p * n)(*[it[1].data_ptr() for it in self.items])
        self.ntaps, self.rows, self.cols, self.rows_pad = [(C.c_int * n)(*[int(it[k]) for it in self.items]) for k in (2, 3, 4, 5)]

    def run(self):
        if self.n:
            check(lib().shm_transpose_taps_multi(self.n, self.w, self.wt, self.ntaps, self.rows, self.cols, self.rows_pad, self.dt, _stream()),
                  "shm_transpose_taps_multi")


def transpose_taps(w, wt, ntaps, rows, cols, rows_pad):
    check(lib().shm_transpose_taps(_p(w), _p(wt), ntaps, rows, cols, rows_pad, _dt(wt), _stream()), "shm_transpose_taps")


def conv2d_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope,
               cin_real=None, gsum=None):
    """cin_real: un-padded input channels, only used for the algorithmic flop count.
    gsum = (aux, ldaux, red): shm_conv2d_fwd_gsum (the stride-2 forward form is Conv2DTranspose's input gradient)."""
    ho, wo = -(-hi // stride), -(-wi // stride)
    flops = 2.0 * batch * ho * wo * ksize * ksize * (cin_real or cin) * cout
    label = f"fwd n{batch} h{hi} {cin}->{cout} k{ksize} s{stride}"
    if gsum is None:
        _timed("", flops, lambda: check(
            lib().shm_conv2d_fwd(_p(x), _p(x2), c1, ldx, ldx2, _p(wk), _p(bias), _p(y), ldy, batch, hi, wi,
                                 cin, cout, ksize, stride, slope, _dtg(x, y), _stream()), "shm_conv2d_fwd"), label)
        return
    aux, ldaux, red = gsum
    _timed("", flops, lambda: check(
        lib().shm_conv2d_fwd_gsum(_p(x), _p(x2), c1, ldx, ldx2, _p(wk), _p(bias), _p(y), ldy, batch, hi, wi, cin, cout, ksize, stride,
                                  slope, _p(aux), ldaux, _p(red), _dtg(x, y), _stream()), "shm_conv2d_fwd_gsum"), label + " +gsum")


STATS_SLOTS = 16            # SHM_STATS_SLOTS


NORM_EXACT, NORM_SCALED = 0, 1          # SHM_NORM_* of include/shmgan_hip.h


def conv2d_in_fwd(x, x2, c1, ldx, ldx2, wk, bias, y, ldy, batch, hi, wi, cin, cout, ksize, stride, slope, stats, eps,
                  cin_real=None, scratch=None, nt_x=None, nt_x2=None, nt_out=None, beta_out=None, norm_mode=NORM_EXACT):
    """conv2d_fwd fused with the InstanceNorm statistics of its output (stats <- mean, inv-std).
    scratch: optional f64 [STATS_SLOTS * batch * cout * 2] (spreads the statistics atomics).
    nt_x / nt_x2: x / x2 is the UN-normalised activation of an InstanceNorm block and this is that block's table
    (shm_conv2d_in_fwd_norm: NORM_EXACT -- the kernel normalises its operand tile in LDS; NORM_SCALED -- wk and bias are
    conv2d_norm_prepare's per-sample operands); nt_out (+ beta_out): this block's own table."""
    ho, wo = -(-hi // stride), -(-wi // stride)
    flops = 2.0 * batch * ho * wo * ksize * ksize * (cin_real or cin) * cout
    label = f"fwd n{batch} h{hi} {cin}->{cout} k{ksize} s{stride}"
    if nt_x is None and nt_x2 is None and nt_out is None:
        _timed("", flops, lambda: check(
            lib().shm_conv2d_in_fwd(_p(x), _p(x2), c1, ldx, ldx2, _p(wk), _p(bias), _p(y), ldy, batch, hi, wi,
                                    cin, cout, ksize, stride, slope, _p(stats), _p(scratch), eps, _dt(x), _stream()),
            "shm_conv2d_in_fwd"), label)
        return
    _timed("", flops, lambda: check(
        lib().shm_conv2d_in_fwd_norm(_p(x), _p(x2), c1, ldx, ldx2, _p(nt_x), _p(nt_x2), norm_mode, _p(wk), _p(bias), _p(y), ldy, batch, hi, wi,
                                     cin, cout, ksize, stride, slope, _p(stats), _p(scratch), eps, _p(nt_out), _p(beta_out), _dt(x), _stream()),
        "shm_conv2d_in_fwd_norm"), label + (" +norm" if (nt_x is not None or nt_x2 is not None) else ""))


def conv2d_norm_prepare(wk, bias, nt, c, part_lo, wk_n, bias_n, batch, cin, cout, ksize):
    """NORM_SCALED operands: per-sample weights (the folded source's channels times inv) and bias rows."""
    check(lib().shm_conv2d_norm_prepare(_p(wk), _p(bias), _p(nt), c, part_lo, _p(wk_n), _p(bias_n), batch, cin, cout, ksize, _dt(wk), _stream()),
          "shm_conv2d_norm_prepare")


def conv2d_wgrad_norm_workspace(batch, hi, wi, cin, cout, ksize, dtype):
    return int(lib().shm_conv2d_wgrad_norm_workspace(batch, hi, wi, cin, cout, ksize, 1 if dtype == torch.bfloat16 else 0))


def conv2d_wgrad_norm_finish(dw, nt, dzsum, batch, c, part_lo, cin, cout, ksize):
    """NORM_SCALED weight gradient, second term: dw[tap][part_lo + k][co] += sum_n (beta - mean * inv)[n][k] * dzsum[n][co]."""
    check(lib().shm_conv2d_wgrad_norm_finish(_p(dw), _p(nt), _p(dzsum), batch, c, part_lo, cin, cout, ksize, _stream()), "shm_conv2d_wgrad_norm_finish")


def in_bwd_keep_dz_sums(dst):
    """The next in_bwd / in_bwd_apply / in_bwd_rank1 call also copies its per-sample channel sums of dz ([batch][c] float64) to dst."""
    check(lib().shm_in_bwd_keep_dz_sums(_p(dst)), "shm_in_bwd_keep_dz_sums")


def conv2d_norm_supported(batch, hi, wi, cin, c1, cout, ksize, stride, norm_part, dtype):
    """Would conv2d_in_fwd(nt_x= / nt_x2=) run on a kernel that normalises source `norm_part` in LDS?  (c1: channels of x when
    there are two sources, else 0; dtype: torch dtype of the activations.)"""
    dt = 1 if dtype == torch.bfloat16 else 0
    return bool(lib().shm_conv2d_norm_supported(batch, hi, wi, cin, c1, cout, ksize, stride, norm_part, dt))


def conv2d_wgrad_norm_supported(batch, hi, wi, cin, cin_ld, c1, cout, ksize, stride, norm_part, dtype):
    dt = 1 if dtype == torch.bfloat16 else 0
    return bool(lib().shm_conv2d_wgrad_norm_supported(batch, hi, wi, cin, cin_ld, c1, cout, ksize, stride, norm_part, dt))


def in_norm_table(stats, beta, nt, batch, c):
    check(lib().shm_in_norm_table(_p(stats), _p(beta), _p(nt), batch, c, _stream()), "shm_in_norm_table")


GSUM_SLOTS = 8              # SHM_GSUM_SLOTS
# InstanceNorm-backward sums in the producing epilogue (model.py).  Default by activation dtype: ON in float32 -- the convolutions
# are MFMA bound there and take the extra aux read in their stride: -3.4 ms of reduce passes for +1.3 ms of epilogues per step at
# BASELINE configs[1] -- OFF in bfloat16, where the same epilogues sit in latency-bound kernels and cost what the reduce passes
# saved (rocprofv3, profiles/README.md round 3).  SHM_GSUM=0 / 1 overrides both (A/B measurements).
import os as _os


def gsum_default(dtype):
    env = _os.environ.get("SHM_GSUM")
    if env is not None:
        return env not in ("0", "")
    return dtype == torch.float32


def conv2d_dgrad(dy, lddy, w, dx, dx2, n1, lddx, lddx2, batch, hi, wi, cin, cout, ksize, stride, gsum=None, gsum2=None):
    """gsum / gsum2 = (aux, ldaux, red) for the dx / dx2 part: the epilogue also delivers the InstanceNorm-backward sums of the
    block whose output gradient that part is (shm_conv2d_dgrad_gsum; red = f64 [GSUM_SLOTS * batch * channels * 2], zero on entry)."""
    ho, wo = -(-hi // stride), -(-wi // stride)
    flops = 2.0 * batch * ho * wo * ksize * ksize * cin * cout
    label = f"dgrad n{batch} h{hi} {cin}<-{cout} k{ksize} s{stride}"
    if gsum is None and gsum2 is None:
        _timed("", flops, lambda: check(
            lib().shm_conv2d_dgrad(_p(dy), lddy, _p(w), _p(dx), _p(dx2), n1, lddx, lddx2, batch, hi, wi, cin,
                                   cout, ksize, stride, _dtg(dy, dx), _stream()), "shm_conv2d_dgrad"), label)
        return
    a1, l1, r1 = gsum or (None, 0, None)
    a2, l2, r2 = gsum2 or (None, 0, None)
    _timed("", flops, lambda: check(
        lib().shm_conv2d_dgrad_gsum(_p(dy), lddy, _p(w), _p(dx), _p(dx2), n1, lddx, lddx2, batch, hi, wi, cin, cout, ksize, stride,
                                    _p(a1), l1, _p(r1), _p(a2), l2, _p(r2), _dtg(dy, dx), _stream()), "shm_conv2d_dgrad_gsum"),
           label + " +gsum")


def conv2d_transpose_fwd(x, ldx, w, bias, y, ldy, batch, hi, wi, cin, cout, slope):
    flops = 2.0 * batch * hi * wi * 9 * cin * cout
    _timed("", flops, lambda: check(
        lib().shm_conv2d_transpose_fwd(_p(x), ldx, _p(w), _p(bias), _p(y), ldy, batch, hi, wi, cin, cout,
                                       slope, _dt(x), _stream()), "shm_conv2d_transpose_fwd"),
           f"convT n{batch} h{hi} {cin}->{cout}")


def conv2d_wgrad_workspace(batch, ho, wo, cin, cout, ksize):
    return int(lib().shm_conv2d_wgrad_workspace(batch, ho, wo, cin, cout, ksize))


def conv2d_wgrad(x, x2, c1, ldx, ldx2, dy, lddy, dw, batch, hi, wi, cin, cin_ld, cout, ksize, stride,
                 accumulate, ws, nt_x=None, nt_x2=None, norm_mode=NORM_EXACT):
    """nt_x / nt_x2: x / x2 is the un-normalised activation of an InstanceNorm block, normalised in LDS (shm_conv2d_wgrad_norm)."""
    ho, wo = -(-hi // stride), -(-wi // stride)
    flops = 2.0 * batch * ho * wo * ksize * ksize * cin * cout
    label = f"wgrad n{batch} h{hi} {cin}x{cout} k{ksize} s{stride}"
    wsb = ws.numel() * ws.element_size()
    if (nt_x is not None or nt_x2 is not None) and TIMER is None:
        check(lib().shm_conv2d_wgrad_norm(_p(x), _p(x2), c1, ldx, ldx2, _p(nt_x), _p(nt_x2), norm_mode, _p(dy), lddy, _p(dw), batch, hi, wi, cin, cin_ld,
                                          cout, ksize, stride, int(accumulate), _p(ws), wsb, _dt(x), _stream()), "shm_conv2d_wgrad_norm")
        return
    if TIMER is None:
        check(lib().shm_conv2d_wgrad(_p(x), _p(x2), c1, ldx, ldx2, _p(dy), lddy, _p(dw), batch, hi, wi, cin, cin_ld,
                                     cout, ksize, stride, int(accumulate), _p(ws), wsb, _dt(x), _stream()),
              "shm_conv2d_wgrad")
        return
    # timing mode: the two phases as separate calls, so the MFMA kernel's events hold nothing else
    import ctypes
    ns = ctypes.c_int(0)
    if nt_x is not None or nt_x2 is not None:
        TIMER.wrap("", flops, lambda: check(
            lib().shm_conv2d_wgrad_partial_norm(_p(x), _p(x2), c1, ldx, ldx2, _p(nt_x), _p(nt_x2), norm_mode, _p(dy), lddy, batch, hi, wi, cin, cin_ld, cout,
                                                ksize, stride, _p(ws), wsb, _dt(x), ctypes.addressof(ns), _stream()),
            "shm_conv2d_wgrad_partial_norm"), label + " +norm")
    else:
        TIMER.wrap("", flops, lambda: check(
            lib().shm_conv2d_wgrad_partial(_p(x), _p(x2), c1, ldx, ldx2, _p(dy), lddy, batch, hi, wi, cin, cin_ld, cout,
                                           ksize, stride, _p(ws), wsb, _dt(x), ctypes.addressof(ns), _stream()),
            "shm_conv2d_wgrad_partial"), label)
    TIMER.wrap("wgrad_reduce_kernel", 0.0, lambda: check(
        lib().shm_conv2d_wgrad_reduce(_p(ws), _p(dw), ksize * ksize * cin * cout, ns.value, int(accumulate), _stream()),
        "shm_conv2d_wgrad_reduce"), label, fixed=True)


def in_stats(a, lda, stats, batch, hw, c, eps):
    check(lib().shm_in_stats(_p(a), lda, _p(stats), batch, hw, c, eps, _dt(a), _stream()), "shm_in_stats")


def _tb(t, n_elems):
    return float(n_elems) * t.element_size()


def in_apply_pool(a, lda, stats, beta, out, ldo, pooled, ldp, batch, h, w, c):
    e = batch * h * w * c                          # read a, write out, write pooled (a quarter)
    _timed_bytes("shm_in_apply_pool", _tb(a, 2.25 * e), lambda: check(
        lib().shm_in_apply_pool(_p(a), lda, _p(stats), _p(beta), _p(out), ldo, _p(pooled), ldp, batch, h, w, c, _dt(a), _stream()),
        "shm_in_apply_pool"))


def in_pool(a, lda, stats, beta, pooled, ldp, batch, h, w, c):
    """pooled = AveragePooling2D(2)(IN apply(a)) without writing the normalised tensor (its other consumers normalise on the fly)."""
    e = batch * h * w * c                          # read a, write pooled (a quarter)
    _timed_bytes("shm_in_pool", _tb(a, 1.25 * e), lambda: check(
        lib().shm_in_pool(_p(a), lda, _p(stats), _p(beta), _p(pooled), ldp, batch, h, w, c, _dt(a), _stream()), "shm_in_pool"))


def in_apply(a, lda, stats, beta, out, ldo, batch, hw, c):
    _timed_bytes("shm_in_apply", _tb(a, 2 * batch * hw * c), lambda: check(
        lib().shm_in_apply(_p(a), lda, _p(stats), _p(beta), _p(out), ldo, batch, hw, c, _dt(a), _stream()), "shm_in_apply"))


def set_abort_words(dev_word, host_word):
    """shm_set_abort_words: dev_word a uint32 / int32 CUDA tensor of one element, host_word a PINNED host tensor of one element (ROCm maps
    pinned host memory into the device's address space at the same address); None, None disarms."""
    if host_word is not None:
        assert host_word.is_pinned() and not host_word.is_cuda and dev_word.is_cuda
    check(lib().shm_set_abort_words(_p(dev_word), _p(host_word)), "shm_set_abort_words")


def set_clock_probe(dev2):
    """shm_set_clock_probe: dev2 = an int64 CUDA tensor of two elements (or None)."""
    check(lib().shm_set_clock_probe(_p(dev2)), "shm_set_clock_probe")


def in_bwd_fused_doubles(batch, hw, c):
    """SHM_IN_BWD_FUSED_DOUBLES: float64 elements of the one-pass form's scratch (per-block partial rows, means, counters and flags)."""
    cb = min(c, 64)
    return (batch * (hw * cb // 16384) * 3 * c + 1) // 2 + batch * c + batch * (c // cb) * 288 + 1


def in_bwd(g1, ldg1, g2, ldg2, a, lda, stats, red, dz, lddz, dbias, batch, h, w, c, slope, fused=None):
    """fused: float64 scratch of in_bwd_fused_doubles(batch, h * w, c) elements (zero on entry, zero on return): the call may run the one-pass
    bf16 form (shm_in_bwd_fused_scratch); the library falls back to reduce + apply on shapes that form does not take."""
    e = batch * h * w * c
    rd = _tb(g1, e * (1.25 if g2 is not None else 1.0)) + _tb(a, e)

    def nb():
        # bytes of the path the call took: the one-pass form reads g1 [+ g2 / 4] and a once and writes dz; reduce + apply read g and a twice
        # (round-5 advisor: three passes were counted for every call, understating the two-pass calls)
        return (rd if last_kernel().startswith("in_bwd_fused") else 2 * rd) + _tb(dz, e)

    def run():
        if fused is not None:
            check(lib().shm_in_bwd_fused_scratch(_p(fused), fused.numel()), "shm_in_bwd_fused_scratch")
        try:
            check(lib().shm_in_bwd(_p(g1), ldg1, _p(g2), ldg2, _p(a), lda, _p(stats), _p(red), _p(dz), lddz, _p(dbias),
                                   batch, h, w, c, slope, _dtg(a, g1), _stream()), "shm_in_bwd")
        finally:
            if fused is not None:               # one-shot state of the library: never left armed behind an error
                lib().shm_in_bwd_fused_scratch(None, 0)
    _timed_bytes("shm_in_bwd", nb, run)


def in_bwd_apply(g1, ldg1, g2, ldg2, a, lda, stats, beta, red, redp, dstage, dz, lddz, dbias, batch, h, w, c, slope):
    """shm_in_bwd without its reduce pass: the sums come from the gsum epilogues of the launches that wrote g1 / g2."""
    e = batch * h * w * c
    nb = _tb(g1, e * (1.25 if g2 is not None else 1.0)) + _tb(a, e) + _tb(dz, e)
    _timed_bytes("shm_in_bwd_apply", nb, lambda: check(
        lib().shm_in_bwd_apply(_p(g1), ldg1, _p(g2), ldg2, _p(a), lda, _p(stats), _p(beta), _p(red), _p(redp), _p(dstage), _p(dz), lddz,
                               _p(dbias), batch, h, w, c, slope, _dtg(a, g1), _stream()), "shm_in_bwd_apply"))


LRELU_RED_SLOTS = 64        # SHM_LRELU_RED_SLOTS


def lrelu_bwd(dy, lddy, y, ldy, dz, lddz, dbias, npix, c, slope, red=None):
    """red: f64 scratch [LRELU_RED_SLOTS * c], required with dbias."""
    check(lib().shm_lrelu_bwd(_p(dy), lddy, _p(y), ldy, _p(dz), lddz, _p(dbias), _p(red), npix, c, slope, _dtg(y, dy), _stream()),
          "shm_lrelu_bwd")


def avgpool2_fwd(x, ldx, y, ldy, batch, h, w, c):
    check(lib().shm_avgpool2_fwd(_p(x), ldx, _p(y), ldy, batch, h, w, c, _dt(x), _stream()), "shm_avgpool2_fwd")


def cvt_f64_f32(src, dst, n, accumulate):
    check(lib().shm_cvt_f64_f32(_p(src), _p(dst), n, int(accumulate), _stream()), "shm_cvt_f64_f32")


def zero(t):
    check(lib().shm_zero(_p(t), t.numel() * t.element_size(), _stream()), "shm_zero")


def head_fwd(x, ldx, w, bias, y, npix, c, slope):
    check(lib().shm_head_fwd(_p(x), ldx, _p(w), _p(bias), _p(y), npix, c, slope, _dt(x), _stream()), "shm_head_fwd")


def head_in_fwd(a, lda, stats, beta, w, bias, y, batch, hw, c, slope):
    check(lib().shm_head_in_fwd(_p(a), lda, _p(stats), _p(beta), _p(w), _p(bias), _p(y), batch, hw, c, slope, _dt(a), _stream()), "shm_head_in_fwd")


def head_in_bwd(a, lda, stats, beta, w, y, dy, dx, lddx, dw_acc, db_acc, batch, hw, c, slope, red, dz_out=None):
    """dx may be None (then dz_out is required): the head's input gradient is dz_out (x) w, see in_bwd_rank1."""
    check(lib().shm_head_in_bwd(_p(a), lda, _p(stats), _p(beta), _p(w), _p(y), _p(dy), _p(dx), lddx, _p(dz_out), _p(dw_acc), _p(db_acc), _p(red), batch,
                                hw, c, slope, _dt(a) if dx is None else _dtg(a, dx), _stream()), "shm_head_in_bwd")


def in_bwd_rank1(hdz, hw_, a, lda, stats, red, dz, lddz, dbias, batch, h, w, c, slope):
    check(lib().shm_in_bwd_rank1(_p(hdz), _p(hw_), _p(a), lda, _p(stats), _p(red), _p(dz), lddz, _p(dbias), batch, h, w, c, slope, _dt(a), _stream()),
          "shm_in_bwd_rank1")


def head_bwd(x, ldx, w, y, dy, dx, lddx, dw_acc, db_acc, npix, c, slope, red=None):
    """red: f64 scratch [LRELU_RED_SLOTS * (c + 1)] (allocated per call when omitted: tests only)."""
    if red is None:
        red = torch.empty(LRELU_RED_SLOTS * (c + 1), dtype=torch.float64, device=x.device)
    check(lib().shm_head_bwd(_p(x), ldx, _p(w), _p(y), _p(dy), _p(dx), lddx, _p(dw_acc), _p(db_acc), _p(red), npix, c,
                             slope, _dtg(x, dx), _stream()), "shm_head_bwd")


def patch_fwd(x, ldx, w, y, batch, h, wd, c, slope):
    check(lib().shm_patch_fwd(_p(x), ldx, _p(w), _p(y), batch, h, wd, c, slope, _dt(x), _stream()), "shm_patch_fwd")


def patch_bwd(x, ldx, w, y, dy, dz, dx, lddx, dw, batch, h, wd, c, slope):
    check(lib().shm_patch_bwd(_p(x), ldx, _p(w), _p(y), _p(dy), _p(dz), _p(dx), lddx, _p(dw), batch, h, wd, c,
                              slope, _dtg(x, dx), _stream()), "shm_patch_bwd")


def dense_fwd(x, w, y, batch, k, nout):
    check(lib().shm_dense_fwd(_p(x), _p(w), _p(y), batch, k, nout, _dt(x), _stream()), "shm_dense_fwd")


def dense_bwd(x, w, dy, dx, dw, batch, k, nout):
    check(lib().shm_dense_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), batch, k, nout, _dtg(x, dx), _stream()), "shm_dense_bwd")


def mul_mask(x, mask, y, n, scale):
    check(lib().shm_mul_mask(_p(x), _p(mask), _p(y), n, scale, _dt(x), _stream()), "shm_mul_mask")


def rgb2yuv_std(rgb, yuv, acc, scale_out, batch, npix):
    check(lib().shm_rgb2yuv_std(_p(rgb), _p(yuv), _p(acc), _p(scale_out), batch, npix, _stream()), "shm_rgb2yuv_std")


def avg_cbcr(ys, out, n):
    check(lib().shm_avg_cbcr(_p(ys[0]), _p(ys[1]), _p(ys[2]), _p(ys[3]), _p(ys[4]), _p(out), n, _stream()),
          "shm_avg_cbcr")


def build_gen_input(ys, gen_y, flags_mask, mode, out, batch, npix):
    check(lib().shm_build_gen_input(_p(ys[0]), _p(ys[1]), _p(ys[2]), _p(ys[3]), _p(ys[4]), _p(gen_y), flags_mask,
                                    mode, _p(out), out.shape[-1], batch, npix, _dt(out), _stream()), "shm_build_gen_input")


def cyc_input_bwd(dcyc, flags_mask, dgen_y, batch, npix):
    check(lib().shm_cyc_input_bwd(_p(dcyc), dcyc.shape[-1], flags_mask, _p(dgen_y), batch, npix, _dt(dcyc), _stream()),
          "shm_cyc_input_bwd")


def yuv2rgb(ych, cbcr, noise, rgb, dpad, nimg, batch, npix):
    check(lib().shm_yuv2rgb(_p(ych), _p(cbcr), _p(noise), _p(rgb), _p(dpad), 0 if dpad is None else dpad.shape[-1], nimg,
                            batch, npix, 0 if dpad is None else _dt(dpad), _stream()), "shm_yuv2rgb")


def pack_rgb16(rgb, noise, dpad, npix_total):
    check(lib().shm_pack_rgb16(_p(rgb), _p(noise), _p(dpad), dpad.shape[-1], npix_total, _dt(dpad), _stream()),
          "shm_pack_rgb16")


def rgb16_to_dy(d16, dy, npix_total, accumulate):
    check(lib().shm_rgb16_to_dy(_p(d16), d16.shape[-1], _p(dy), npix_total, int(accumulate), _dt(d16), _stream()),
          "shm_rgb16_to_dy")


def randn(out, stddev, seed, stream_id=0):
    """out ~ N(0, stddev^2), Philox-4x32-10 keyed by (seed, stream_id)."""
    check(lib().shm_randn(_p(out), out.numel(), stddev, int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id), _stream()), "shm_randn")


def keep_mask(out, rate, seed, stream_id=0):
    """out = 1 with probability 1 - rate else 0 (Dropout keep mask)."""
    check(lib().shm_keep_mask(_p(out), out.numel(), rate, int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id), _stream()), "shm_keep_mask")


XENT_TF_FUSED, XENT_INTENDED = 0, 1          # SHM_XENT_* of include/shmgan_hip.h


def dhead_losses(rf, cls, loss, drf_d, dcls_d, drf_g, batch, np_, target, xent_mode=XENT_TF_FUSED):
    """xent_mode: XENT_TF_FUSED = the class-logit gradient TF's fused softmax-cross-entropy kernel returns (softmax - labels:
    the reference as executed), XENT_INTENDED = the true derivative for the un-normalised D1 label row."""
    check(lib().shm_dhead_losses(_p(rf), _p(cls), _p(loss), _p(drf_d), _p(dcls_d), _p(drf_g), batch, np_, target,
                                 int(xent_mode), _stream()), "shm_dhead_losses")


def image_losses_workspace(batch, s):
    return int(lib().shm_image_losses_workspace(batch, s))


def image_losses(gen_rgb, cyc_rgb, cyc_y, cbcr, orig_ptrs, ds_ptrs, flags_mask, style_factor, loss, dgen_y, dcyc_y,
                 ws, batch, s):
    """orig_ptrs / ds_ptrs: ctypes arrays of 5 device pointers (host memory, read at call time)."""
    check(lib().shm_image_losses(_p(gen_rgb), _p(cyc_rgb), _p(cyc_y), _p(cbcr), orig_ptrs, ds_ptrs, flags_mask,
                                 style_factor, _p(loss), _p(dgen_y), _p(dcyc_y), _p(ws),
                                 ws.numel() * ws.element_size(), batch, s, _stream()), "shm_image_losses")


def adam_clip(w, m, v, g, n, alpha, beta1, beta2, eps, gscale):
    _timed_bytes("shm_adam_clip", 7.0 * 4 * n, lambda: check(          # read w, m, v, g; write w, m, v
        lib().shm_adam_clip(_p(w), _p(m), _p(v), _p(g), n, alpha, beta1, beta2, eps, gscale, _stream()), "shm_adam_clip"))


# ---- SpecSeg (inference only) ----------------------------------------------------------------
def pack_channels(src, ldsrc, c0, nc, dst, lddst, npix):
    check(lib().shm_pack_channels(_p(src), ldsrc, c0, nc, _p(dst), lddst, npix, _stream()), "shm_pack_channels")


def bn_apply(a, lda, gamma, beta, mean, var, eps, out, ldo, npix, c):
    check(lib().shm_bn_apply(_p(a), lda, _p(gamma), _p(beta), _p(mean), _p(var), eps, _p(out), ldo, npix, c,
                             _stream()), "shm_bn_apply")


def maxpool2_fwd(x, ldx, y, ldy, batch, h, w, c):
    check(lib().shm_maxpool2_fwd(_p(x), ldx, _p(y), ldy, batch, h, w, c, _stream()), "shm_maxpool2_fwd")


def conv2d_transpose2x2_fwd(x, ldx, w, bias, y, ldy, batch, hi, wi, cin, cout, slope=1.0):
    flops = 2.0 * batch * hi * wi * 4 * cin * cout
    _timed("", flops, lambda: check(
        lib().shm_conv2d_transpose2x2_fwd(_p(x), ldx, _p(w), _p(bias), _p(y), ldy, batch, hi, wi, cin, cout,
                                          slope, _dt(x), _stream()), "shm_conv2d_transpose2x2_fwd"),
           f"convT2 n{batch} h{hi} {cin}->{cout}")


def head_sigmoid_fwd(x, ldx, w, bias, y, npix, c):
    check(lib().shm_head_sigmoid_fwd(_p(x), ldx, _p(w), _p(bias), _p(y), npix, c, _stream()), "shm_head_sigmoid_fwd")


def spec_loss(cyc_y, cbcr, ds_ptrs, mask, loss, batch, npix):
    check(lib().shm_spec_loss(_p(cyc_y), _p(cbcr), ds_ptrs, _p(mask), _p(loss), batch, npix, _stream()),
          "shm_spec_loss")


# ---- live attention branch --------------------------------------------------------------------------
def mask_pool_pack(mask, dst, batch, s, k):
    """MaxPooling2D(k) of mask [batch,s,s,1] into channel 0 of the activation tensor dst [batch,s/k,s/k,ld]."""
    check(lib().shm_mask_pool_pack(_p(mask), _p(dst), dst.shape[-1], batch, s, k, _dt(dst), _stream()), "shm_mask_pool_pack")


def add_bcast(a, b, out, nimg, per, nb, i0=0):
    check(lib().shm_add_bcast(_p(a), _p(b), _p(out), nimg, per, nb, i0, _dt(a), _stream()), "shm_add_bcast")


def sum_groups(src, dst, nimg, per, nb, i0=0, accumulate=False):
    check(lib().shm_sum_groups(_p(src), _p(dst), nimg, per, nb, i0, int(accumulate), _dt(src), _stream()), "shm_sum_groups")


# ---- input pipeline ------------------------------------------------------------------------------
def resize_bilinear_u8(src_u8, dst, scale=1.0 / 255.0, flip_ud=False):
    """src_u8 [hin,win,c] uint8 device tensor -> dst [ho,wo,c] float32 (tf.image.resize bilinear, then * scale)."""
    hin, win, c = src_u8.shape
    ho, wo, _ = dst.shape
    check(lib().shm_resize_bilinear_u8(_p(src_u8), hin, win, c, _p(dst), ho, wo, scale, int(flip_ud), _stream()),
          "shm_resize_bilinear_u8")


# ---- first-layer input gradient, summed over input channels -----------------------------------------
def sum_input_channels(w, cin, cout, mask, weff):
    check(lib().shm_sum_input_channels(_p(w), cin, cout, mask, _p(weff), _stream()), "shm_sum_input_channels")


def conv3x3_dgrad_sum1(dz, lddz, weff, out, nk, batch, hi, wi, c, stride, accumulate):
    check(lib().shm_conv3x3_dgrad_sum1(_p(dz), lddz, _p(weff), _p(out), nk, batch, hi, wi, c, stride, int(accumulate),
                                       _dt(dz), _stream()), "shm_conv3x3_dgrad_sum1")
