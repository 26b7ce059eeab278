"""SHMGAN generator (U-Net) and discriminator (PatchGAN + classifier) on the HIP kernels.

Mirrors `build_generator` (SHM.py:228-327) and `build_discriminator` (SHM.py:343-389) of
/root/reference/ShmGANwithSSpecSeg.py "as executed" (SURVEY.md findings 3-5): block order
Conv -> bias -> LeakyReLU(0.2) -> InstanceNormalization(eps=1e-6, gamma=1, constant beta);
the SpecSeg attention term is a constant zero added to the skips; Concatenate puts the
upsampled tensor first.  Forward AND backward are explicit sequences of C-ABI calls
(shmgan_amd.ops); torch tensors are storage.

Weights live in one flat fp32 buffer per model (so clip+Adam is one launch and the
data-parallel all-reduce is one contiguous bucket); `trainable_variables` are views into
it in the reference's Keras order and layouts (HWIO; Conv2DTranspose [kh,kw,Cout,Cin];
Dense [in,out]).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops

LRELU = 0.2
IN_EPS = 1e-6
# The fused block: the InstanceNorm apply of a block whose consumers stage their operand as an LDS halo image can be done by those
# consumers (ops.conv2d_in_fwd(nt_x=) / ops.conv2d_wgrad(nt_x=)); the normalised tensor is then never written.  Results are the
# same bits either way, so where to fold is a matter of measured time (tools/bench_fold.py):
#   "auto" (default)  fold where it pays: float32 blocks whose consumer is the weights-in-registers kernel (one source of at most
#                     64 channels, i.e. the 256 x 256 level at filter_size 64: the pass it removes costs 265 us per n = 40 tensor,
#                     the consumers' extra work -20 .. +50 us).  On the halo kernels the in-LDS pass costs about what the stand-alone
#                     pass does at 128 x 128 and more below; in bfloat16, whose MFMA loops are 6 x shorter, it costs 2-4 x the pass.
#   "all"             fold wherever the kernels can (tests, A/B measurements)
#   "0"               never
_FOLD_VALUES = {"0": False, "": False, "1": "auto", "auto": "auto", "all": "all"}
if os.environ.get("SHM_NORM_FOLD", "auto") not in _FOLD_VALUES:
    raise ValueError(f"SHM_NORM_FOLD={os.environ['SHM_NORM_FOLD']!r}: expected one of auto, all, 0")
NORM_FOLD = _FOLD_VALUES[os.environ.get("SHM_NORM_FOLD", "auto")]
# How a folding consumer normalises: "exact" (SHM_NORM_EXACT, default) in LDS, bit-identical to the stand-alone pass; "scaled"
# (SHM_NORM_SCALED) in the operands -- per-sample weights w * inv and bias rows, the weight gradient's slabs scaled per sample plus a
# rank-n term from the per-sample dz sums.  Same result to rounding; measured slower than "exact" where "exact" pays and no faster
# elsewhere, so it is an option, not the default.
if os.environ.get("SHM_NORM_MODE", "exact") not in ("exact", "scaled"):
    raise ValueError(f"SHM_NORM_MODE={os.environ['SHM_NORM_MODE']!r}: expected exact or scaled")
NORM_MODE = ops.NORM_SCALED if os.environ.get("SHM_NORM_MODE", "exact") == "scaled" else ops.NORM_EXACT
WGRAD_AFTER_DGRAD = os.environ.get("SHM_WGRAD_AFTER_DGRAD", "0") == "1"
PAD_C = 16          # channel pitch of 3- and 10-channel images in float32 (one 64-byte MFMA staging row)


def pad_channels(dtype):
    """Channel pitch of the 3- and 10-channel images = contraction granule of the tap GEMM:
    one 64-byte LDS row, i.e. 16 float32 or 32 bfloat16 channels."""
    return 64 // torch.empty((), dtype=dtype).element_size()


def _padk(c, g):
    return (c + g - 1) // g * g


class Arena:
    """Named, shape-keyed tensor cache: every activation buffer is allocated once and reused
    on the next step (the step has a static memory plan; 288 GB of HBM is not the limit)."""

    def __init__(self, device):
        self.device = device
        self.t = {}

    def get(self, name, shape, dtype=torch.float32):
        key = (name, tuple(shape), dtype)
        t = self.t.get(key)
        if t is None:
            # f64 buffers are kernel scratch (slot copies of statistics etc.): some are "zero on entry, zero on
            # return" by contract, so they start out zeroed
            alloc = torch.zeros if dtype == torch.float64 else torch.empty
            t = alloc(tuple(shape), dtype=dtype, device=self.device)
            if dtype == torch.float64 and torch.device(self.device).type == "cuda":
                # the zero fill runs on whichever stream is current (a part of a two-stream forward allocates on the second
                # stream) while slices of the buffer are about to be used on another one: wait for it once, at allocation
                torch.cuda.current_stream(self.device).synchronize()
            self.t[key] = t
        return t

    def get_slack(self, name, shape, dtype, slack):
        """A tensor of `shape` at the front of a zero-filled flat allocation with `slack` more elements behind it: the compact 3-channel
        images (one 16-byte chunk per pixel) are also read by the generic conv kernels under a forced variant, which fetch a whole
        K-channel operand row per pixel -- the last pixels' rows then end inside the slack instead of past the allocation (kernels that
        address through buffer descriptors are bounded anyway; the flat-address ones are not: round-4 advisor finding)."""
        key = (name, tuple(shape), dtype, int(slack))
        t = self.t.get(key)
        if t is None:
            n = 1
            for d in shape:
                n *= int(d)
            flat = torch.zeros(n + int(slack), dtype=dtype, device=self.device)
            if torch.device(self.device).type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()
            t = flat[:n].view(tuple(shape))
            self.t[key] = t
            self.t[key + ("storage",)] = flat
        return t

    def fused_timeouts(self, clear=True):
        """Names of the one-pass InstanceNorm-backward scratches (_fused_scratch) whose timeout word is set: a sample barrier of
        in_bwd_fused8_kernel gave up waiting (the launch then went on with wrong means instead of hanging).  One device-to-host copy.
        The words are u32 flags in the last float64 slot: compared as integers (a 1 read as a float64 is a denormal, and flush-to-zero
        anywhere on the way would hide it: round-5 advisor) and cleared once reported, so the scratch is "zero on entry" again."""
        items = [(k[0], t) for k, t in self.t.items() if isinstance(k[0], str) and k[0].startswith("bwd/fused/")]
        if not items:
            return []
        words = torch.stack([t[-1:].view(torch.int64)[0] for _, t in items]).cpu().tolist()
        hit = [(name, t) for (name, t), w in zip(items, words) if w != 0]
        if clear:
            for _, t in hit:
                t[-1:].zero_()
        return [name for name, _ in hit]

    def nbytes(self):
        # a get_slack tensor is a view of its "storage" entry: count the storage
        return sum(t.numel() * t.element_size() for k, t in self.t.items() if not (len(k) == 4 and isinstance(k[3], int)))


def _fused_scratch(arena, adt, n, hw, c):
    """Scratch of the one-pass bfloat16 InstanceNorm backward (ops.in_bwd(fused=)): zero-filled float64, left zero by every call."""
    if adt != torch.bfloat16:
        return None
    return arena.get(f"bwd/fused/{n}x{hw}x{c}", (ops.in_bwd_fused_doubles(n, hw, c),), torch.float64)


class WgradLane:
    """Second HIP stream for the weight-gradient launches.  In a backward pass only the
    in_bwd -> dgrad chain is on the critical path; every wgrad just needs its layer's dz.  Issuing
    them on their own stream lets the hardware fill the tail of each (dependent) kernel on the main
    stream with wgrad workgroups.  Buffers a wgrad reads (dz per layer, forward activations) are not
    rewritten before `join()`."""

    def __init__(self, device, enabled=True):
        self.stream = torch.cuda.Stream(device=device) if enabled else None

    def submit(self, fn):
        if self.stream is None:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record()                                   # everything issued so far on the main stream
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            fn()

    def event(self):
        """Event marking the completion of everything submitted so far (None if disabled)."""
        if self.stream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def join(self):
        ev = self.event()
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)


class _ConvTurns:
    """Turn taking of two half-batch passes that run on two streams: before(p) makes part p's stream wait for the other part's
    most recent convolution, after(p) marks part p's convolution.  The host issues the parts alternately, one convolution (and the
    elementwise passes behind it) at a time, so an event is always recorded before the other part waits on it."""

    def __init__(self, streams):
        self.streams = streams
        self.ev = [None, None]

    def before(self, p):
        ev = self.ev[1 - p]
        if ev is not None:
            self.streams[p].wait_event(ev)

    def after(self, p):
        ev = torch.cuda.Event()
        ev.record(self.streams[p])
        self.ev[p] = ev


def generator_layers(f):
    """(keras_name, kind, k, cin, cout) in Keras creation order (Generator_summary.txt)."""
    return [
        ("conv2d", "c", 3, 10, f), ("conv2d_1", "c", 3, f, f),
        ("conv2d_4", "c", 3, f, 2 * f), ("conv2d_5", "c", 3, 2 * f, 2 * f),
        ("conv2d_8", "c", 3, 2 * f, 4 * f), ("conv2d_9", "c", 3, 4 * f, 4 * f),
        ("conv2d_12", "c", 3, 4 * f, 8 * f), ("conv2d_13", "c", 3, 8 * f, 8 * f),
        ("conv2d_16", "c", 1, 8 * f, 8 * f), ("conv2d_17", "c", 1, 8 * f, 8 * f),
        ("conv2d_transpose", "t", 3, 8 * f, 8 * f),
        ("conv2d_18", "c", 3, 16 * f, 8 * f), ("conv2d_19", "c", 3, 8 * f, 8 * f),
        ("conv2d_transpose_1", "t", 3, 8 * f, 4 * f),
        ("conv2d_20", "c", 3, 8 * f, 4 * f), ("conv2d_21", "c", 3, 4 * f, 4 * f),
        ("conv2d_transpose_2", "t", 3, 4 * f, 2 * f),
        ("conv2d_22", "c", 3, 4 * f, 2 * f), ("conv2d_23", "c", 3, 2 * f, 2 * f),
        ("conv2d_transpose_3", "t", 3, 2 * f, f),
        ("conv2d_24", "c", 3, 2 * f, f), ("conv2d_25", "c", 3, f, f),
        ("conv2d_26", "c", 1, f, 1),
    ]


class _Vars:
    """Flat parameter / gradient / Adam-moment storage with named views."""

    def __init__(self, shapes, order, device):
        """shapes: list of shapes in Keras variable order; order: storage order (indices)."""
        self.shapes = [tuple(s) for s in shapes]
        sizes = [int(np.prod(s)) for s in self.shapes]
        self.n = sum(sizes)
        self.offsets = [0] * len(shapes)
        off = 0
        for i in order:
            self.offsets[i] = off
            off += sizes[i]
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.m = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.v = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.vars = [self.flat[o:o + z].view(s) for o, z, s in zip(self.offsets, sizes, self.shapes)]
        self.grads = [self.grad[o:o + z].view(s) for o, z, s in zip(self.offsets, sizes, self.shapes)]
        self.iterations = 0
        self._sizes = sizes
        self.op_flat, self.op_vars = self.flat, self.vars       # MFMA operand copy (== master in float32)

    def operand_copy(self, dtype):
        """bf16 path: the products that read weights 'as stored' (dgrad, Conv2DTranspose) take a bf16
        copy of the whole flat buffer, refreshed once per step by one cast kernel."""
        if dtype != torch.float32:
            self.op_flat = torch.zeros(self.n, dtype=dtype, device=self.flat.device)
            self.op_vars = [self.op_flat[o:o + z].view(s) for o, z, s in zip(self.offsets, self._sizes, self.shapes)]

    def refresh_operands(self):
        if self.op_flat is not self.flat:
            ops.cast_f32(self.flat, self.op_flat, self.n)

    def load(self, arrays):
        assert len(arrays) == len(self.vars)
        for v, a in zip(self.vars, arrays):
            a = torch.as_tensor(np.asarray(a, dtype=np.float32))
            assert tuple(a.shape) == tuple(v.shape), (a.shape, v.shape)
            v.copy_(a)

    def numpy(self, which="vars"):
        return [t.detach().cpu().numpy().copy() for t in getattr(self, which)]


class _ModelBase:
    trainable = True

    def count_params(self):
        return self.P.n

    # storage index of every variable in Keras creation order (identity unless the live attention branch adds variables,
    # which Keras creates in the middle of the layer list)
    keras_index = None

    def _korder(self):
        return self.keras_index if self.keras_index is not None else list(range(len(self.P.vars)))

    @property
    def trainable_variables(self):
        return [self.P.vars[i] for i in self._korder()]

    @property
    def gradients(self):
        return [self.P.grads[i] for i in self._korder()]

    def get_weights(self):
        w = self.P.numpy("vars")
        return [w[i] for i in self._korder()]

    def set_weights(self, arrays):
        ko = self._korder()
        assert len(arrays) == len(ko)
        st = [None] * len(ko)
        for a, i in zip(arrays, ko):
            st[i] = a
        self.P.load(st)
        self.weights_dirty = True


class AttentionBranch:
    """One `attention_layer` call (SHM.py:404-412) on the LIVE SpecSeg mask of the step (attention="live";
    as executed the reference only ever evaluates it on a constant zero mask, SURVEY finding 3):
    MaxPooling2D(k) -> Conv2D(1->C, 3x3, bias, LeakyReLU) -> Conv2D(C->C, 3x3, bias, LeakyReLU) on [B, S/k, S/k].
    The variables (k1 [3,3,1,C], b1, k2 [3,3,C,C], b2) are views into the owning model's flat parameter buffer, so
    clip+Adam and the gradient all-reduce cover them; forward and backward are C-ABI calls like every other layer."""

    def __init__(self, owner, vi, c, pool, tag):
        self.o, self.vi, self.c, self.pool, self.tag = owner, vi, c, pool, tag
        self.wk1 = torch.zeros(9 * c * owner.pad, dtype=owner.adt, device=owner.dev)
        self.wk2 = torch.zeros(9 * c * c, dtype=owner.adt, device=owner.dev)
        self.ctx = None

    def transpose_items(self):
        P, c = self.o.P, self.c
        return [(P.vars[self.vi], self.wk1, 9, 1, c, self.o.pad), (P.vars[self.vi + 2], self.wk2, 9, c, c, c)]

    def prepare_weights(self):
        for it in self.transpose_items():
            ops.transpose_taps(*it)

    def forward(self, mask, B, S):
        """mask [B,S,S,1] fp32 -> attention map [B, S/pool, S/pool, C] (activation dtype)."""
        o, c, A = self.o, self.c, self.o.arena
        h = S // self.pool
        m = A.get(f"{self.tag}/m", (B, h, h, o.pad), o.adt)
        y1 = A.get(f"{self.tag}/y1", (B, h, h, c), o.adt)
        y2 = A.get(f"{self.tag}/y2", (B, h, h, c), o.adt)
        ops.mask_pool_pack(mask, m, B, S, self.pool)
        ops.conv2d_fwd(m, None, 0, o.pad, 0, self.wk1, o.P.vars[self.vi + 1], y1, c, B, h, h, o.pad, c, 3, 1, LRELU, cin_real=1)
        ops.conv2d_fwd(y1, None, 0, c, 0, self.wk2, o.P.vars[self.vi + 3], y2, c, B, h, h, c, c, 3, 1, LRELU)
        self.ctx = dict(B=B, h=h, m=m, y1=y1, y2=y2)
        return y2

    def backward(self, dattn):
        """dattn [B,h,h,C] ([G]-typed): gradient at the attention map, summed over every copy of the sample.  Fills the
        weight gradients (kernels by the MFMA wgrad, biases through the owner's f64 accumulators)."""
        o, c, A, x = self.o, self.c, self.o.arena, self.ctx
        B, h = x["B"], x["h"]
        dz2 = A.get(f"{self.tag}/dz2", (B, h, h, c), o.adt)
        dz1 = A.get(f"{self.tag}/dz1", (B, h, h, c), o.adt)
        dy1 = A.get(f"{self.tag}/dy1", (B, h, h, c), o.gdt)
        red = A.get(f"attn/lred/{c}", (ops.LRELU_RED_SLOTS * c,), torch.float64)
        ops.lrelu_bwd(dattn, c, x["y2"], c, dz2, c, o._acc_slice(self.vi + 3), B * h * h, c, LRELU, red)
        ws = o.ws_provider(ops.conv2d_wgrad_workspace(B, h, h, c, c, 3))
        o.lane.submit(lambda: ops.conv2d_wgrad(x["y1"], None, 0, c, 0, dz2, c, o.P.grads[self.vi + 2], B, h, h, c, c, c, 3, 1, 0, ws))
        ops.conv2d_dgrad(dz2, c, o.P.op_vars[self.vi + 2], dy1, None, c, c, 0, B, h, h, c, c, 3, 1)
        ops.lrelu_bwd(dy1, c, x["y1"], c, dz1, c, o._acc_slice(self.vi + 1), B * h * h, c, LRELU, red)
        ws = o.ws_provider(ops.conv2d_wgrad_workspace(B, h, h, 1, c, 3))
        o.lane.submit(lambda: ops.conv2d_wgrad(x["m"], None, 0, o.pad, 0, dz1, c, o.P.grads[self.vi], B, h, h, 1, o.pad, c, 3, 1, 0, ws))


# =============================================================================== Generator
class Generator(_ModelBase):
    name = "SHM_Generator"

    def __init__(self, image_size, filter_size, device, arena, ws_provider, lane=None, dtype=torch.float32,
                 grad_dtype=None, attention=False):
        self.S, self.F, self.dev = image_size, filter_size, device
        self.attention = bool(attention)
        self.arena, self.ws_provider = arena, ws_provider
        self.lane = lane or WgradLane(device, enabled=False)
        self.adt = dtype                                   # activation / MFMA operand dtype
        self.gdt = (grad_dtype or dtype) if dtype != torch.float32 else torch.float32   # gradient-signal tensors ([G] in the header)
        self.pad = pad_channels(dtype)
        self.gsum = ops.gsum_default(dtype)                # InstanceNorm-backward sums in the producing epilogues
        self.fold = NORM_FOLD                              # InstanceNorm apply folded into the consumers: False / "auto" / "all" (see NORM_FOLD)
        self.norm_mode = NORM_MODE                         # ... in LDS (exact) or in the operands (scaled)
        self._plans = {}
        assert image_size % 16 == 0 and filter_size % self.pad == 0, \
            f"image_size must be a multiple of 16 and filter_size of {self.pad}"
        self.layers = generator_layers(filter_size)
        shapes = []
        for _, kind, k, cin, cout in self.layers:
            shapes.append((k, k, cin, cout) if kind == "c" else (k, k, cout, cin))
            shapes.append((cout,))
        nl = len(self.layers)
        nbase = 2 * nl
        akern, abias = [], []
        if self.attention:            # attention_layer variables of the four levels: [k 1->C, b, k C->C, b], C = F, 2F, 4F, 8F
            for lvl in range(4):
                c = filter_size << lvl
                shapes += [(3, 3, 1, c), (c,), (3, 3, c, c), (c,)]
                akern += [nbase + 4 * lvl, nbase + 4 * lvl + 2]
                abias += [nbase + 4 * lvl + 1, nbase + 4 * lvl + 3]
            # Keras creates an attention_layer right after the two convolutions of its encoder level (SHM.py:244-275)
            ko = []
            for lvl in range(4):
                ko += [4 * lvl, 4 * lvl + 1, 4 * lvl + 2, 4 * lvl + 3] + [nbase + 4 * lvl + j for j in range(4)]
            self.keras_index = ko + list(range(16, nbase))
        # storage order: all MFMA-wgrad kernels (the layers', then the attention branch's), then [head kernel, every bias]
        # (f64-accumulated grads)
        order = [2 * i for i in range(nl - 1)] + akern + [2 * (nl - 1)] + [2 * i + 1 for i in range(nl)] + abias
        self.P = _Vars(shapes, order, device)
        self.P.operand_copy(dtype)
        self.acc_off = self.P.offsets[2 * (nl - 1)]            # start of the f64-accumulated region
        self.attn_off = self.P.offsets[akern[0]] if akern else self.acc_off
        self.acc_n = self.P.n - self.acc_off
        self.acc = torch.zeros(self.acc_n, dtype=torch.float64, device=device)
        self.in_channels = [cout for n_, kind, k, cin, cout in self.layers if kind == "c" and n_ != "conv2d_26"]
        self.betas = [torch.zeros(c, dtype=torch.float32, device=device) for c in self.in_channels]
        # K-contiguous copies: conv fwd needs [t][cout][cin_pad]; convT dgrad needs [t][cin][cout]
        self.wk = {}
        for i, (_, kind, k, cin, cout) in enumerate(self.layers[:-1]):
            if kind == "c":
                self.wk[i] = torch.zeros(k * k * cout * _padk(cin, self.pad), dtype=dtype, device=device)
            else:
                self.wk[i] = torch.zeros(9 * cin * cout, dtype=dtype, device=device)
        self.attn = [AttentionBranch(self, nbase + 4 * lvl, filter_size << lvl, 1 << lvl, f"g/attn{lvl}") for lvl in range(4)] \
            if self.attention else []
        self.weights_dirty = True
        self._tb = None
        self.ctx = {}
        self.debug = None
        self._on_wgrad = None
        self._dattn = [None] * 4           # attention skip gradients of the step in flight (reset by zero_grad)

    # -- parameter plumbing ---------------------------------------------------------------
    def set_betas(self, arrays):
        for b, a in zip(self.betas, arrays):
            b.copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)))

    def grad_buckets(self):
        """Data-parallel exchange plan of the flat gradient: [(trigger_layer, [(lo, hi), ...]), ...] in the order the
        backward pass completes them.  The kernels are stored in layer order, and a backward pass finishes the layers
        last to first, so a bucket = the kernels of a run of layers, ready as soon as the weight gradient of its LOWEST
        layer has been issued (`trigger_layer`); the last entry (trigger None) is what only the end of the pass completes:
        the first encoder layers plus the f64-accumulated region (head kernel, every bias).  Stage cuts: decoder top
        (3.7 MB at F=64), decoder bottom (49.6 MB), encoder bottom + 1x1 bottleneck (19.8 MB), the rest (1.1 MB)."""
        off = [self.P.offsets[2 * i] for i in range(len(self.layers))]
        cuts = [16, 10, 4]                                   # lowest layer of each early bucket
        out, hi = [], self.attn_off                          # (live attention kernels sit between the layers' and the head's)
        for c in cuts:
            out.append((c, [(off[c], hi)]))
            hi = off[c]
        out.append((None, [(0, hi), (self.attn_off, self.P.n)]))
        return out

    def _acc_slice(self, var_index):
        o = self.P.offsets[var_index] - self.acc_off
        return self.acc[o:o + self.P.vars[var_index].numel()]

    def prepare_weights(self):
        """Refresh the K-contiguous weight copies after an optimizer step."""
        if not self.weights_dirty:
            return
        self.P.refresh_operands()
        if self._tb is None:              # one launch for every layer's K-contiguous copy
            items = []
            for i, (_, kind, k, cin, cout) in enumerate(self.layers[:-1]):
                w = self.P.vars[2 * i]
                if kind == "c":      # HWIO [t][cin][cout] -> [t][cout][cin_pad]
                    items.append((w, self.wk[i], k * k, cin, cout, _padk(cin, self.pad)))
                else:                # Keras convT [t][cout][cin] -> [t][cin][cout] (for its dgrad)
                    items.append((w, self.wk[i], 9, cout, cin, cout))
            for br in self.attn:
                items += br.transpose_items()
            self._tb = ops.TransposeBatch(items)
        self._tb.run()
        self.weights_dirty = False

    def zero_grad(self):
        """Start of a step: clear the flat gradient and the f64 accumulators, and forget the attention skip gradients of
        the previous step (a step that aborted between its two backward passes must not leak them into this one)."""
        ops.zero(self.P.grad)
        ops.zero(self.acc)
        self._dattn = [None] * 4

    # -- live attention branch ---------------------------------------------------------------
    def attention_forward(self, mask, B):
        """attn_1..attn_4 (SHM.py:248,257,266,275) of the step's SpecSeg mask [B,S,S,1]; pass the result to forward(attn=)."""
        self.prepare_weights()
        self._attn_B = B
        return [br.forward(mask, B, self.S) for br in self.attn]

    def attention_masks(self):
        """Sign patterns of the eight attention LeakyReLUs of the last attention_forward (test diagnostics, see lrelu_masks)."""
        return [(br.ctx[k] > 0).cpu().numpy() for br in self.attn for k in ("y1", "y2")]

    def attention_backward(self):
        """Backward of the four attention branches from the skip gradients every backward() of this step accumulated."""
        for lvl, br in enumerate(self.attn):
            br.backward(self._dattn[lvl])
        self._dattn = [None] * 4

    def finish_grads(self):
        """Fold the f64-accumulated head-kernel / bias gradients into the flat fp32 gradient."""
        ops.cvt_f64_f32(self.acc, self.P.grad[self.acc_off:], self.acc_n, 0)

    # -- forward --------------------------------------------------------------------------
    def _fold_plan(self, nb, n, attn):
        """{layer index of an InstanceNorm block: True} for the blocks whose normalisation is applied by their consumers: every
        convolution that reads the block's output, and that convolution's weight gradient, must run on a kernel that normalises its
        operand in LDS for this batch (the launcher's variant choice depends on it: ops.conv2d_norm_supported).  Encoder block 1 of a
        level feeds the next block; block 2 feeds the pool (ops.in_pool) and, as the skip, the decoder's Concatenate convolution
        (its second source); the Concatenate block feeds the decoder level's second block.  That block's own output goes to a
        Conv2DTranspose (or the head, which has normalised on the fly since round 2) and the bottleneck is 1x1: not folded.
        Live attention adds its maps to the normalised skips, which therefore have to exist: nothing is folded."""
        key = (nb, n, bool(attn), self.fold)            # nb: samples per forward launch (a part of a two-part forward), n: per backward launch
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        plan = {}
        if self.fold and not attn:
            dt, L = self.adt, self.layers

            def ok(h, cin, c1, cout, part):
                cin_p = _padk(cin, self.pad)
                if self.fold != "all" and not (dt == torch.float32 and c1 == 0 and cin_p <= 64):
                    return False                               # "auto": only where it is measured to pay
                return (ops.conv2d_norm_supported(nb, h, h, cin_p, c1, cout, 3, 1, part, dt) and
                        ops.conv2d_wgrad_norm_supported(n, h, h, cin, cin_p, c1, cout, 3, 1, part, dt))
            for lvl in range(4):
                h = self.S >> lvl
                li0, li1 = 2 * lvl, 2 * lvl + 1
                plan[li0] = ok(h, L[li0][4], 0, L[li1][4], 0)
                lic = 11 + 3 * (3 - lvl)                       # the decoder level's Concatenate convolution: sources [u, skip]
                cu = L[lic - 1][4]
                plan[li1] = ok(h, L[lic][3], cu, L[lic][4], 1)
                plan[lic] = ok(h, L[lic][4], 0, L[lic + 1][4], 0)
        self._plans[key] = plan
        return plan

    def _cnl_fwd(self, tag, li, bi, x, x2, c1, ldx, ldx2, n, h, w, r0, r1, part, pooled=None, apply=True, sync=None, ntx=None, ntx2=None,
                 fold=False):
        """Conv2D(k, s1, bias, LeakyReLU) -> InstanceNormalization on the samples [r0, r1) of a batch of n.  x, x2, pooled are
        FULL-batch tensors (the record describes them: the backward pass runs on the whole batch); the launches take row views.
        Returns (ahat, record).  pooled: AveragePooling2D(2) of the result, written by the same pass as the normalisation.
        apply=False: the consumer normalises on the fly (the head, ops.head_in_fwd): returns the un-normalised tensor.
        ntx / ntx2: x / x2 is the un-normalised output of a folded block and this is its table.  fold: this block's output is
        normalised by its consumers -- returns the un-normalised tensor, record["nt"] is the table they need."""
        _, _, k, cin, cout = self.layers[li]
        cin_p = _padk(cin, self.pad)
        A = self.arena
        nb = r1 - r0
        a = A.get(f"{tag}/a{li}", (n, h, w, cout), self.adt)
        ahat = A.get(f"{tag}/h{li}", (n, h, w, cout), self.adt) if apply and not fold else None
        nt = A.get(f"{tag}/nt{li}", (n, 4, cout), torch.float32) if fold else None
        stats = A.get(f"{tag}/s{li}", (n * cout * 2,), torch.float64)
        st = stats[r0 * cout * 2:r1 * cout * 2]
        # zero-on-return scratch: one per concurrently running part
        scr = A.get(f"stats_scratch/{nb * cout}/p{part}", (ops.STATS_SLOTS * nb * cout * 2,), torch.float64)
        if sync is not None:
            sync.before(part)
        wk_, bias_, mode = self.wk[li], self.P.vars[2 * li + 1], ops.NORM_EXACT
        folded_in = (ntx is not None or ntx2 is not None)
        if folded_in and self.norm_mode == ops.NORM_SCALED:
            # the normalisation in the operands: one weight copy (the folded source's channels times inv) and one bias row per sample
            src_nt, lo, c = (ntx, 0, ldx) if ntx is not None else (ntx2, c1, ldx2)
            wn = A.get(f"{tag}/wn{li}", (n, k * k * cout * cin_p), self.adt)
            bn = A.get(f"{tag}/bn{li}", (n, cout), torch.float32)
            ops.conv2d_norm_prepare(wk_, bias_, src_nt[r0:r1], c, lo, wn[r0:r1], bn[r0:r1], nb, cin_p, cout, k)
            wk_, bias_, mode = wn[r0:r1], bn[r0:r1], ops.NORM_SCALED
        ops.conv2d_in_fwd(x[r0:r1], None if x2 is None else x2[r0:r1], c1, ldx, ldx2, wk_, bias_, a[r0:r1], cout, nb, h, w,
                          cin_p, cout, k, 1, LRELU, st, IN_EPS, cin_real=cin, scratch=scr,
                          nt_x=None if ntx is None else ntx[r0:r1], nt_x2=None if ntx2 is None else ntx2[r0:r1],
                          nt_out=None if nt is None else nt[r0:r1], beta_out=self.betas[bi] if fold else None, norm_mode=mode)
        if sync is not None:
            sync.after(part)
        rec = dict(li=li, x=x, x2=x2, c1=c1, ldx=ldx, ldx2=ldx2, a=a, stats=stats, h=h, w=w, bi=bi, n=n, cout=cout, ntx=ntx, ntx2=ntx2, nt=nt)
        if not apply:
            return a, rec
        if fold:
            if pooled is not None:
                ops.in_pool(a[r0:r1], cout, st, self.betas[bi], pooled[r0:r1], cout, nb, h, w, cout)
            return a, rec
        if pooled is not None:
            ops.in_apply_pool(a[r0:r1], cout, st, self.betas[bi], ahat[r0:r1], cout, pooled[r0:r1], cout, nb, h, w, cout)
        else:
            ops.in_apply(a[r0:r1], cout, st, self.betas[bi], ahat[r0:r1], cout, nb, h * w, cout)
        return ahat, rec

    def forward(self, x16, tag, attn=None, parts=1):
        """x16: [N,S,S,pad] (10 real channels, zero padded to the 64-byte pitch).  Returns gen_Y [N,S,S,1].
        attn: attention_forward()'s maps ([B,...], N a multiple of B: image i is a copy of sample i % B): added to the four
        skip tensors, `down_k + attn_k` (SHM.py:290-293); the pooled path keeps the un-augmented tensor.
        parts = 2 (experiment, not the trainer's default): the batch is evaluated as two halves, the second one on the second
        stream, taking turns convolution by convolution (samples are independent: InstanceNorm is per sample, so this is exact).
        A forward pass is a dependency chain conv -> statistics -> normalise -> conv with nothing beside it (rocprofv3 timeline:
        2 ms of a 121 ms fp32 step are normalisation passes with no MFMA kernel running); the idea was to run one half's
        normalisation pass under the other half's convolution.  Measured: +2 ms -- the halves' convolutions no longer overlap
        and the half-size grids are less efficient than the hidden passes are long."""
        n, S = x16.shape[0], self.S
        assert tuple(x16.shape) == (n, S, S, self.pad) and x16.dtype == self.adt
        self.prepare_weights()
        if parts == 2 and self.lane.stream is not None and n >= 4 and (attn is None or (n // 2) % self._attn_B == 0):
            # two half batches, ALTERNATING: a half launches its next convolution only when the other half's previous one has
            # finished (events), so the halves do not fall into lockstep (conv beside conv, normalisation beside normalisation):
            # half A's normalisation pass runs under half B's convolution and vice versa
            cut = n // 2
            main, second = torch.cuda.current_stream(), self.lane.stream
            start = torch.cuda.Event()
            start.record(main)
            second.wait_event(start)
            sync = _ConvTurns((main, second))
            gens = [self._forward_rows(x16, tag, attn, 0, cut, 0, sync), self._forward_rows(x16, tag, attn, cut, n, 1, sync)]
            live = [True, True]
            y = None
            while live[0] or live[1]:
                for p in (0, 1):
                    if live[p]:
                        with torch.cuda.stream(sync.streams[p]):
                            try:
                                next(gens[p])
                            except StopIteration as e:
                                live[p] = False
                                y = e.value if p == 0 else y
            self.lane.join()
            return y
        gen = self._forward_rows(x16, tag, attn, 0, n, 0, None)
        while True:
            try:
                next(gen)
            except StopIteration as e:
                return e.value

    def _forward_rows(self, x16, tag, attn, r0, r1, part, sync):
        """Generator: one step = one convolution of the samples [r0, r1) with the elementwise passes behind it."""
        n, S, F = x16.shape[0], self.S, self.F
        nb = r1 - r0
        A = self.arena
        recs = []
        plan = self._fold_plan(nb, n, attn is not None)
        cur, cur_nt, ld, h = x16, None, self.pad, S      # cur_nt: cur is the un-normalised output of a folded block, with this table
        li = bi = 0
        downs = []
        for lvl in range(4):
            pooled = None
            for j in range(2):
                if j == 1:            # the level's second block: its normalisation pass also writes the pooled tensor
                    pooled = A.get(f"{tag}/p{lvl}", (n, h // 2, h // 2, self.layers[li][4]), self.adt)
                cur, r = self._cnl_fwd(tag, li, bi, cur, None, 0, ld, 0, n, h, h, r0, r1, part, pooled=pooled, sync=sync, ntx=cur_nt,
                                       fold=plan.get(li, False))
                cur_nt = r["nt"]
                r["pooled"] = pooled
                recs.append(r)
                yield
                ld = self.layers[li][4]
                li += 1
                bi += 1
            if attn is not None:
                skip = A.get(f"{tag}/skip{lvl}", (n, h, h, ld), self.adt)
                ops.add_bcast(cur[r0:r1], attn[lvl], skip[r0:r1], nb, h * h * ld, self._attn_B, r0)
                downs.append((skip, ld, h, None))
            else:
                downs.append((cur, ld, h, cur_nt))
            cur, cur_nt, h = pooled, None, h // 2
        for _ in range(2):                       # the two 1x1 blocks
            cur, r = self._cnl_fwd(tag, li, bi, cur, None, 0, ld, 0, n, h, h, r0, r1, part, sync=sync)
            recs.append(r)
            li += 1
            bi += 1
            yield
        ups = []
        for lvl in range(4):
            _, _, _, cin, cout = self.layers[li]
            u = A.get(f"{tag}/u{lvl}", (n, 2 * h, 2 * h, cout), self.adt)
            if sync is not None:
                sync.before(part)
            ops.conv2d_transpose_fwd(cur[r0:r1], ld, self.P.op_vars[2 * li], self.P.vars[2 * li + 1], u[r0:r1], cout, nb, h, h, cin,
                                     cout, LRELU)
            if sync is not None:
                sync.after(part)
            yield
            ups.append(dict(li=li, x=cur, ldx=ld, u=u, h=h))
            li += 1
            h *= 2
            skip, sld, sh, skip_nt = downs[3 - lvl]
            assert sh == h
            cur, r = self._cnl_fwd(tag, li, bi, u, skip, cout, cout, sld, n, h, h, r0, r1, part, sync=sync, ntx2=skip_nt,
                                   fold=plan.get(li, False))                                                  # concat [u, skip]
            cur_nt = r["nt"]
            recs.append(r)
            ld = self.layers[li][4]
            li += 1
            bi += 1
            yield
            # the last block's InstanceNorm is applied by the head kernels (forward and backward) on the fly
            cur, r = self._cnl_fwd(tag, li, bi, cur, None, 0, ld, 0, n, h, h, r0, r1, part, apply=lvl < 3, sync=sync, ntx=cur_nt)
            recs.append(r)
            li += 1
            bi += 1
            yield
        y = A.get(f"{tag}/y", (n, S, S, 1))
        ops.head_in_fwd(cur[r0:r1], ld, r["stats"][r0 * ld * 2:r1 * ld * 2], self.betas[r["bi"]], self.P.vars[2 * li], self.P.vars[2 * li + 1], y[r0:r1],
                        nb, S * S, ld, LRELU)
        if part == 0:                 # the records describe full-batch tensors: identical whichever part builds them
            self.ctx[tag] = dict(n=n, recs=recs, ups=ups, head_x=cur, head_rec=r, y=y, x16=x16, attn=attn is not None)
        return y

    # -- backward -------------------------------------------------------------------------
    def _gsum(self, rec, pooled=False):
        """(aux, ldaux, red) for the launch that writes the gradient at `rec`'s InstanceNorm output (its epilogue then delivers the
        sums of that block's InstanceNorm backward, ops.conv2d_dgrad(gsum=)).  pooled: the launch writes the gradient of the
        AveragePooling2D output instead; its sums go against the pooled normalised tensor (the next level's input)."""
        if not self.gsum:
            return None
        n, c = rec["n"], rec["cout"]
        key = "gredp" if pooled else "gred"
        red = self.arena.get(f"bwd/{key}/L{rec['li']}/{n}", (ops.GSUM_SLOTS * n * c * 2,), torch.float64)
        rec[key] = red
        return (rec["pooled"] if pooled else rec["a"]), c, red

    def _cnl_bwd(self, tag, rec, g1, g2, n, need_dx, dx=None, dx2=None, n1=0, rank1=None, gsum=None, gsum2=None):
        """Backward of one Conv->LReLU->IN block.  g1: gradient at the IN output (same res),
        g2: optional gradient of the 2x2 average pool that consumed the IN output.
        rank1 = (hdz, w): g1 is the rank-1 tensor hdz (x) w of the head (formed on the fly, g1 = None).
        gsum / gsum2: _gsum() of the block(s) whose output gradient dx / dx2 is.  If the launches that wrote g1 (and g2) were given
        this block's _gsum(), the InstanceNorm backward is the single apply pass (ops.in_bwd_apply), otherwise reduce + apply.
        Accumulates dW / dbias; returns nothing (dx/dx2 are written if need_dx)."""
        li, h, w = rec["li"], rec["h"], rec["w"]
        _, _, k, cin, cout = self.layers[li]
        A = self.arena
        dz = A.get(f"bwd/dz/L{li}/{n}", (n, h, w, cout), self.adt)       # per layer: read later by the wgrad lane
        # a source folded in the operands (SHM_NORM_SCALED): the weight gradient's second term needs the per-sample channel sums of dz,
        # which the InstanceNorm backward below stages on the way to the bias gradient
        scaled = self.norm_mode == ops.NORM_SCALED and (rec["ntx"] is not None or rec["ntx2"] is not None)
        dzsum = None
        if scaled:
            dzsum = A.get(f"bwd/dzsum/L{li}/{n}", (n, cout), torch.float64)
            ops.in_bwd_keep_dz_sums(dzsum)
        try:
            if rank1 is not None:
                red = A.get(f"bwd/red/{n * cout}", (n * cout * 3,), torch.float64)
                ops.in_bwd_rank1(rank1[0], rank1[1], rec["a"], cout, rec["stats"], red, dz, cout, self._acc_slice(2 * li + 1), n, h, w, cout, LRELU)
            elif rec.get("gred") is not None and (g2 is None) == (rec.get("gredp") is None):
                dstage = A.get(f"bwd/dstage/{n * cout}", (n * cout,), torch.float64)
                ops.in_bwd_apply(g1, cout, g2, cout, rec["a"], cout, rec["stats"], self.betas[rec["bi"]], rec.pop("gred"), rec.pop("gredp", None),
                                 dstage, dz, cout, self._acc_slice(2 * li + 1), n, h, w, cout, LRELU)
            else:
                red = A.get(f"bwd/red/{n * cout}", (n * cout * 3,), torch.float64)
                ops.in_bwd(g1, cout, g2, cout, rec["a"], cout, rec["stats"], red, dz, cout, self._acc_slice(2 * li + 1), n,
                           h, w, cout, LRELU, fused=_fused_scratch(A, self.adt, n, h * w, cout))
        finally:
            # the request is one-shot, thread-local state of the library, consumed by the call above; if Python raised before reaching
            # it (an arena allocation, a bad argument) it must not stay armed for an unrelated InstanceNorm backward (advisor, round 3)
            if scaled:
                ops.in_bwd_keep_dz_sums(None)
        if self.debug is not None:           # test diagnostics: keep the per-layer gradients
            if g1 is None:
                g1 = rank1[0].reshape(n, h, w, 1) * rank1[1].reshape(1, 1, 1, -1)
            self.debug[li] = (g1.clone(), None if g2 is None else g2.clone(), dz.clone())
        cin_p = _padk(cin, self.pad)
        ws = self.ws_provider(ops.conv2d_wgrad_norm_workspace(n, h, w, cin, cout, k, self.adt) if scaled else ops.conv2d_wgrad_workspace(n, h, w, cin, cout, k))

        def wgrad_launches():
            ops.conv2d_wgrad(rec["x"], rec["x2"], rec["c1"], rec["ldx"], rec["ldx2"], dz, cout, self.P.grads[2 * li], n, h, w, cin, cin_p, cout, k, 1, 1, ws,
                             nt_x=rec["ntx"], nt_x2=rec["ntx2"],
                             norm_mode=ops.NORM_SCALED if scaled else ops.NORM_EXACT)
            if scaled:
                src_nt, lo, c = (rec["ntx"], 0, rec["ldx"]) if rec["ntx"] is not None else (rec["ntx2"], rec["c1"], rec["ldx2"])
                ops.conv2d_wgrad_norm_finish(self.P.grads[2 * li], src_nt, dzsum, n, c, lo, cin, cout, k)

        def wgrad():
            self.lane.submit(wgrad_launches)
            if self._on_wgrad is not None:
                self._on_wgrad(li)
        # WGRAD_AFTER_DGRAD: the weight gradient is released behind the layer's input gradient instead of beside it.  Both are MFMA
        # bound -- side by side they only share the matrix pipes -- whereas the NEXT thing on the main stream is the InstanceNorm
        # backward of the layer below, an HBM-bound pass that then runs under this weight gradient instead of alone
        if not WGRAD_AFTER_DGRAD:
            wgrad()
        if need_dx:
            lddx = rec["ldx"]
            ops.conv2d_dgrad(dz, cout, self.P.op_vars[2 * li], dx, dx2, n1, lddx, rec["ldx2"], n, h, w, cin, cout, k, 1,
                             gsum=gsum, gsum2=gsum2)
        if WGRAD_AFTER_DGRAD:
            wgrad()
        return dz

    def backward(self, dy, tag, need_dx=False, on_wgrad=None):
        """dy: gradient wrt gen_Y [N,S,S,1].  Accumulates into the flat gradient (+ f64 region).
        on_wgrad(li): called right after the weight gradient of layer li has been issued on the wgrad lane (the
        data-parallel trainer launches the gradient bucket that layer completes).
        need_dx=True: returns the gradient wrt the padded input (channels 0..9 valid).
        need_dx="dz": returns the first layer's pre-activation gradient dz [N,S,S,F] instead -- the caller only
        needs channel sums of the input gradient and forms them with ops.conv3x3_dgrad_sum1 (no 64->10 dgrad)."""
        c = self.ctx[tag]
        n, recs, ups = c["n"], c["recs"], c["ups"]
        A = self.arena
        S, F = self.S, self.F
        nl = len(self.layers)
        self._on_wgrad = on_wgrad
        try:
            return self._backward(dy, c, n, recs, ups, A, S, F, nl, tag, need_dx)
        finally:
            self._on_wgrad = None

    def _backward(self, dy, c, n, recs, ups, A, S, F, nl, tag, need_dx):
        # head
        # head: weight / bias gradients and the scalar factor hdz of its input gradient (hdz (x) w is never written: the
        # backward of the block in front of the head forms it on the fly)
        hx = c["head_x"]
        hdz = A.get(f"bwd/hdz/{n}x{S}", (n, S, S), torch.float32)
        hred = A.get(f"bwd/hred/{F}", (ops.LRELU_RED_SLOTS * (F + 1),), torch.float64)
        hr = c["head_rec"]
        ops.head_in_bwd(hx, F, hr["stats"], self.betas[hr["bi"]], self.P.vars[2 * (nl - 1)], c["y"], dy, None, 0, self._acc_slice(2 * (nl - 1)),
                        self._acc_slice(2 * (nl - 1) + 1), n, S * S, F, LRELU, hred, dz_out=hdz)
        dcur = None
        ri = len(recs) - 1
        dskips = [None] * 4
        for lvl in range(3, -1, -1):
            up = ups[lvl]
            r2, r1 = recs[ri], recs[ri - 1]
            ri -= 2
            h = r2["h"]
            cout = self.layers[r2["li"]][4]
            dmid = A.get(f"bwd/dm/{n}x{h}x{cout}", (n, h, h, cout), self.gdt)
            self._cnl_bwd(tag, r2, dcur, None, n, True, dmid, None, cout,
                          rank1=(hdz, self.P.vars[2 * (nl - 1)].reshape(-1)) if lvl == 3 else None, gsum=self._gsum(r1))
            # concat block: split gradient into (du, dskip); dskip is the gradient at the encoder block's InstanceNorm output
            # (live attention adds its map to the skip: the same gradient)
            cu = r1["c1"]
            cs = self.layers[r1["li"]][3] - cu
            du = A.get(f"bwd/du/{n}x{h}x{cu}", (n, h, h, cu), self.gdt)
            dsk = A.get(f"bwd/dskip{3 - lvl}/{n}x{h}x{cs}", (n, h, h, cs), self.gdt)
            enc2 = recs[2 * (3 - lvl) + 1]                           # second block of encoder level 3 - lvl
            self._cnl_bwd(tag, r1, dmid, None, n, True, du, dsk, cu, gsum2=self._gsum(enc2))
            dskips[3 - lvl] = dsk
            if c["attn"]:                       # d attn_k = sum of the skip gradient over every copy of the sample
                B = self._attn_B                # (state reset by zero_grad() at the start of every step)
                da = A.get(f"bwd/dattn{3 - lvl}/{B}", (B, h, h, cs), self.gdt)
                ops.sum_groups(dsk, da, n, h * h * cs, B, 0, accumulate=self._dattn[3 - lvl] is not None)
                self._dattn[3 - lvl] = da
            # Conv2DTranspose: LeakyReLU', bias grad, wgrad (roles swapped), dgrad = stride-2 conv
            tli = up["li"]
            _, _, _, tcin, tcout = self.layers[tli]
            dzu = A.get(f"bwd/dzu/L{tli}/{n}", (n, h, h, cu), self.adt)
            lred = A.get(f"bwd/lred/{cu}", (ops.LRELU_RED_SLOTS * cu,), torch.float64)
            ops.lrelu_bwd(du, cu, up["u"], cu, dzu, cu, self._acc_slice(2 * tli + 1), n * h * h, cu, LRELU, lred)
            ws = self.ws_provider(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, tcout, tcin, 3))
            self.lane.submit(lambda dzu=dzu, up=up, tli=tli, tcout=tcout, tcin=tcin, h=h, ws=ws: ops.conv2d_wgrad(
                dzu, None, 0, tcout, 0, up["x"], up["ldx"], self.P.grads[2 * tli], n, h, h, tcout, tcout, tcin, 3, 2, 1, ws))
            if self._on_wgrad is not None:
                self._on_wgrad(tli)
            hin = up["h"]
            dcur = A.get(f"bwd/d/{n}x{hin}x{tcin}", (n, hin, hin, tcin), self.gdt)
            # input gradient of the Conv2DTranspose = the stride-2 forward form; it writes the gradient at the InstanceNorm output
            # of the block below (recs[ri] after the two decrements above: the previous level's second block, or the bottleneck's)
            ops.conv2d_fwd(dzu, None, 0, tcout, 0, self.wk[tli], None, dcur, tcin, n, h, h, tcout, tcin, 3, 2, 1.0, gsum=self._gsum(recs[ri]))
        # bottleneck 1x1 blocks
        for j in range(2):
            r = recs[ri]
            ri -= 1
            h = r["h"]
            cin = self.layers[r["li"]][3]
            dn = A.get(f"bwd/db{ri}/{n}x{h}x{cin}", (n, h, h, cin), self.gdt)
            # the first 1x1 block's input gradient is the gradient of pool 4: sums against the pooled tensor of encoder level 3
            self._cnl_bwd(tag, r, dcur, None, n, True, dn, None, cin, gsum=self._gsum(recs[ri]) if j == 0 else self._gsum(recs[ri], pooled=True))
            dcur = dn
        dpool = dcur                                   # gradient wrt pool4 output
        for lvl in range(3, -1, -1):
            r2, r1 = recs[ri], recs[ri - 1]
            ri -= 2
            h = r2["h"]
            cout = self.layers[r2["li"]][4]
            dmid = A.get(f"bwd/dm/{n}x{h}x{cout}", (n, h, h, cout), self.gdt)
            self._cnl_bwd(tag, r2, dskips[lvl], dpool, n, True, dmid, None, cout, gsum=self._gsum(r1))
            cin = self.layers[r1["li"]][3]
            if lvl > 0:
                dpool = A.get(f"bwd/dp/{n}x{h}x{cin}", (n, h, h, cin), self.gdt)
                self._cnl_bwd(tag, r1, dmid, None, n, True, dpool, None, cin, gsum=self._gsum(recs[ri], pooled=True))
            else:
                if need_dx is True:
                    dx16 = A.get(f"bwd/dx16/{n}", (n, h, h, self.pad), self.gdt)
                    self._cnl_bwd(tag, r1, dmid, None, n, True, dx16, None, cin)
                    return dx16
                dz0 = self._cnl_bwd(tag, r1, dmid, None, n, False)
                return dz0 if need_dx == "dz" else None
        return None

    def lrelu_tensors(self, tag):
        """The 23 stored LeakyReLU outputs of the forward tagged `tag`, in layer order: the tensors whose sign the
        backward pass reads as the LeakyReLU mask (test diagnostics)."""
        c = self.ctx[tag]
        out = [None] * len(self.layers)
        for r in c["recs"]:
            out[r["li"]] = r["a"]
        for u in c["ups"]:
            out[u["li"]] = u["u"]
        out[-1] = c["y"]
        return out

    def lrelu_masks(self, tag):
        """Sign pattern (a > 0) of the 23 LeakyReLU outputs of the forward tagged `tag`, in layer
        order (test diagnostics: lets the float64 oracle take the same side of every kink)."""
        return [(t > 0).cpu().numpy() for t in self.lrelu_tensors(tag)]

    def __call__(self, x, training=False, tag="call"):
        """Keras-style call on a [N,S,S,10] tensor (reference: self.G(x, training=...))."""
        n = x.shape[0]
        x16 = self.arena.get(f"{tag}/x16", (n, self.S, self.S, self.pad), self.adt)
        x16.zero_()
        x16[..., :10].copy_(x)
        return self.forward(x16, tag)

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        for i, (name, kind, k, cin, cout) in enumerate(self.layers):
            cnt = self.P.vars[2 * i].numel() + self.P.vars[2 * i + 1].numel()
            print_fn(f"{name:24s} {'Conv2DTranspose' if kind == 't' else 'Conv2D':16s} k={k} {cin}->{cout}  {cnt}")
        print_fn(f"Total params: {self.P.n:,}")




# =========================================================================== Discriminator
class Discriminator(_ModelBase):
    name = "SHM_Discriminator"

    def __init__(self, image_size, filter_size, device, arena, ws_provider, dropout=0.2, lane=None, dtype=torch.float32,
                 grad_dtype=None, attention=False):
        self.S, self.F, self.dev = image_size, filter_size, device
        self.attention = bool(attention)
        self.arena, self.ws_provider = arena, ws_provider
        self.lane = lane or WgradLane(device, enabled=False)
        self.adt = dtype
        self.gdt = (grad_dtype or dtype) if dtype != torch.float32 else torch.float32
        self.pad = pad_channels(dtype)
        # pitch of the 3-channel input images (elements): ONE 16-byte chunk per pixel -- r, g, b, 0 in float32; r, g, b and five zeros in
        # bfloat16 -- instead of the 64-byte MFMA staging row (round 4: the padded tensor was 403 MB written and read twice per step for
        # 38-75 MB of pixels).  The first layer's forward and weight gradient have kernels for this layout (csrc: conv3x3s2_rgb_*); the
        # generic kernels also read it correctly (the channels they see beyond the pixel's own are its neighbours', against zero weights)
        self.in_pitch = 4 if dtype == torch.float32 else 8
        if os.environ.get("SHM_D_INPUT") == "staging":       # A/B against the round-3 layout (tools/README.md)
            self.in_pitch = self.pad
        self.gsum = ops.gsum_default(dtype)
        self.dropout = dropout
        assert image_size % 32 == 0
        f, s = filter_size, image_size // 32
        self.s = s
        self.chan = [3, f, 2 * f, 4 * f, 8 * f, 16 * f]
        shapes = [(3, 3, self.chan[i], self.chan[i + 1]) for i in range(5)]
        shapes += [(3, 3, 16 * f, 1), (s * s * 16 * f, 5)]
        self.names = ["conv2d_27", "conv2d_28", "conv2d_29", "conv2d_30", "conv2d_33", "conv2d_34", "dense"]
        order = list(range(7))
        if self.attention:            # attention_layer(8F, pool 16x16) (SHM.py:358): [k 1->8F, b, k 8F->8F, b], biases stored last
            c = 8 * f
            shapes += [(3, 3, 1, c), (c,), (3, 3, c, c), (c,)]
            order += [7, 9, 8, 10]
            self.names += ["conv2d_31", "conv2d_31/bias", "conv2d_32", "conv2d_32/bias"]
            self.keras_index = [0, 1, 2, 3, 7, 8, 9, 10, 4, 5, 6]          # created between the 4th and 5th block
        self.P = _Vars(shapes, order, device)
        self.P.operand_copy(dtype)
        self.acc_off = self.P.offsets[8] if self.attention else self.P.n   # f64-accumulated region: the two attention biases
        self.acc = torch.zeros(self.P.n - self.acc_off, dtype=torch.float64, device=device)
        self.attn = AttentionBranch(self, 7, 8 * f, 16, "d/attn") if self.attention else None
        self.betas = [torch.zeros(c, dtype=torch.float32, device=device) for c in self.chan[1:]]
        self.wk = [torch.zeros(9 * self.chan[i + 1] * _padk(self.chan[i], self.pad), dtype=dtype, device=device)
                   for i in range(5)]
        self.weights_dirty = True
        self._tb = None
        self.ctx = None

    def set_betas(self, arrays):
        for b, a in zip(self.betas, arrays):
            b.copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)))

    def prepare_weights(self):
        if not self.weights_dirty:
            return
        self.P.refresh_operands()
        if self._tb is None:
            items = [(self.P.vars[i], self.wk[i], 9, self.chan[i], self.chan[i + 1], _padk(self.chan[i], self.pad)) for i in range(5)]
            if self.attn is not None:
                items += self.attn.transpose_items()
            self._tb = ops.TransposeBatch(items)
        self._tb.run()
        self.weights_dirty = False

    def zero_grad(self):
        ops.zero(self.P.grad)
        if self.attn is not None:
            ops.zero(self.acc)

    def _acc_slice(self, var_index):
        o = self.P.offsets[var_index] - self.acc_off
        return self.acc[o:o + self.P.vars[var_index].numel()]

    def attention_forward(self, mask, B):
        """attn_disc (SHM.py:358) of the step's SpecSeg mask [B,S,S,1]; pass the result to forward(attn=)."""
        self.prepare_weights()
        self._attn_B = B
        return self.attn.forward(mask, B, self.S)

    def forward(self, xd16, keep_mask=None, mask_rows=(), parts=None, attn=None, join=None):
        """xd16 [N,S,S,16] (rgb + zeros).  keep_mask [len(mask_rows)...] is the Dropout keep mask of
        the `training=True` samples: mask_rows = list of (row_start, nrows, mask_row_start).
        parts: optional list of (row0, row1, run) -- the conv trunk is evaluated per row range (samples
        are independent: InstanceNorm) through `run(fn)`; the trainer uses it to push the real-image
        half onto the second stream while the generator is still producing the fake half; join() is then
        called before the heads."""
        n = xd16.shape[0]
        self.start(xd16, keep_mask, mask_rows, attn)
        if parts is None:
            parts = [(0, n, lambda fn: fn())]
        for r0, r1, run in parts:
            run(lambda r0=r0, r1=r1: self.trunk_rows(r0, r1))
        if join is not None:          # parts that ran on another stream: the heads below read every row
            join()
        return self.heads()

    def start(self, xd16, keep_mask=None, mask_rows=(), attn=None):
        """Declare the batch of the next forward (buffers only); follow with trunk_rows(...) over
        every row and heads().  attn: attention_forward()'s map [B,...] (batch a multiple of B, image i = a copy of sample
        i % B): added after the fourth block, `x + attn_disc` (SHM.py:359)."""
        n, S = xd16.shape[0], self.S
        self.prepare_weights()
        A = self.arena
        bufs = []
        h = S
        for i in range(5):
            cout = self.chan[i + 1]
            ho = h // 2
            bufs.append((A.get(f"d/a{i}/{n}", (n, ho, ho, cout), self.adt), A.get(f"d/h{i}/{n}", (n, ho, ho, cout), self.adt),
                         A.get(f"d/s{i}/{n}", (n * cout * 2,), torch.float64)))
            h = ho
        x3p = A.get(f"d/x3p/{n}", (n, S // 16, S // 16, self.chan[4]), self.adt) if attn is not None else None
        self._pending = dict(n=n, xd16=xd16, bufs=bufs, keep_mask=keep_mask, mask_rows=mask_rows, attn=attn, x3p=x3p)

    def trunk_rows(self, r0, r1):
        """The five Conv(3x3, s2) -> LeakyReLU -> InstanceNorm blocks on samples [r0, r1)."""
        xd16, bufs = self._pending["xd16"], self._pending["bufs"]
        nb = r1 - r0
        cur, ld, h = xd16[r0:r1], xd16.shape[-1], self.S
        for i in range(5):
            cin, cout = self.chan[i], self.chan[i + 1]
            ho = h // 2
            a, ahat, stats = bufs[i]
            st = stats[r0 * cout * 2:r1 * cout * 2]
            # keyed by the row range: two parts evaluated on two streams must not share a zero-on-entry scratch
            scr = self.arena.get(f"d/stats_scratch/{r0}:{r1}/{cout}", (ops.STATS_SLOTS * nb * cout * 2,), torch.float64)
            ops.conv2d_in_fwd(cur, None, 0, ld, 0, self.wk[i], None, a[r0:r1], cout, nb, h, h, _padk(cin, self.pad), cout, 3, 2,
                              LRELU, st, IN_EPS, cin_real=cin, scratch=scr)
            ops.in_apply(a[r0:r1], cout, st, self.betas[i], ahat[r0:r1], cout, nb, ho * ho, cout)
            cur, ld, h = ahat[r0:r1], cout, ho
            if i == 3 and self._pending["attn"] is not None:
                x3p = self._pending["x3p"]
                ops.add_bcast(cur, self._pending["attn"], x3p[r0:r1], nb, ho * ho * cout, self._attn_B, r0)
                cur = x3p[r0:r1]

    def heads(self):
        """Dropout (training=True rows), PatchGAN logits and the Dense(5) classifier on the whole batch."""
        pd = self._pending
        n, xd16, bufs = pd["n"], pd["xd16"], pd["bufs"]
        keep_mask, mask_rows = pd["keep_mask"], pd["mask_rows"]
        A = self.arena
        recs = []
        cur, ld, h = xd16, xd16.shape[-1], self.S
        for i in range(5):
            a, ahat, stats = bufs[i]
            recs.append(dict(x=cur, ldx=ld, a=a, stats=stats, h=h))
            cur, ld, h = ahat, self.chan[i + 1], h // 2
            if i == 3 and pd["x3p"] is not None:
                cur = pd["x3p"]
        c5 = self.chan[5]
        per = h * h * c5
        scale = 1.0 / (1.0 - self.dropout)
        for r0, nr, m0 in mask_rows:                 # Dropout on the training=True samples, in place
            ops.mul_mask(cur[r0:r0 + nr], keep_mask[m0:m0 + nr], cur[r0:r0 + nr], nr * per, scale)
        rf = A.get(f"d/rf/{n}", (n, h, h, 1))
        cls = A.get(f"d/cls/{n}", (n, 5))
        ops.patch_fwd(cur, c5, self.P.vars[5], rf, n, h, h, c5, LRELU)
        ops.dense_fwd(cur, self.P.vars[6], cls, n, per, 5)
        self.ctx = dict(n=n, recs=recs, x5=cur, rf=rf, cls=cls, keep_mask=keep_mask, mask_rows=mask_rows, attn=pd["attn"] is not None)
        return rf, cls

    def _backward(self, n, drf, dcls, params, need_dx):
        """Backward over the first n samples of the last forward.  params: accumulate weight
        gradients; need_dx: return the gradient wrt the 16-pitch input."""
        c = self.ctx
        A = self.arena
        s, c5 = self.s, self.chan[5]
        per = s * s * c5
        x5 = c["x5"]
        dz_p = A.get(f"d/bwd/dzp/{n}", (n, s, s, 1))
        dx5 = A.get(f"d/bwd/dx5/{n}", (n, s, s, c5), self.gdt)
        ops.patch_bwd(x5, c5, self.P.vars[5], c["rf"], drf, dz_p, dx5, c5, self.P.grads[5] if params else None, n, s, s,
                      c5, LRELU)
        if dcls is not None:
            ops.dense_bwd(x5, self.P.vars[6], dcls, dx5, self.P.grads[6] if params else None, n, per, 5)
        scale = 1.0 / (1.0 - self.dropout)
        for r0, nr, m0 in c["mask_rows"]:
            if r0 + nr <= n:
                ops.mul_mask(dx5[r0:r0 + nr], c["keep_mask"][m0:m0 + nr], dx5[r0:r0 + nr], nr * per, scale)
        dcur = dx5
        gred = None
        for i in range(4, -1, -1):
            rec = c["recs"][i]
            cin, cout = self.chan[i], self.chan[i + 1]
            h = rec["h"]
            ho = h // 2
            dz = A.get(f"d/bwd/dz{i}/{n}", (n, ho, ho, cout), self.adt)
            if gred is not None:           # the stride-2 input gradient below delivered this block's sums: one pass instead of two
                ops.in_bwd_apply(dcur, cout, None, 0, rec["a"][:n], cout, rec["stats"], self.betas[i], gred, None, None, dz, cout, None, n, ho, ho,
                                 cout, LRELU)
            else:
                red = A.get(f"d/bwd/red{i}/{n}", (n * cout * 3,), torch.float64)
                ops.in_bwd(dcur, cout, None, 0, rec["a"], cout, rec["stats"], red, dz, cout, None, n, ho, ho, cout, LRELU,
                           fused=_fused_scratch(A, self.adt, n, ho * ho, cout))
            gred = None
            if params:
                ws = self.ws_provider(ops.conv2d_wgrad_workspace(n, ho, ho, cin, cout, 3))
                self.lane.submit(lambda rec=rec, dz=dz, i=i, h=h, cin=cin, cout=cout, ws=ws: ops.conv2d_wgrad(
                    rec["x"], None, 0, rec["ldx"], 0, dz, cout, self.P.grads[i], n, h, h, cin, _padk(cin, self.pad), cout, 3, 2, 0, ws))
            if i == 0 and need_dx == "dz":
                return dz
            if i > 0 or need_dx:
                ldx = rec["ldx"]
                # i == 0: the image gradient in the compact pitch (3 channels in a 4- / 8-element pixel).  The input-gradient kernels write the
                # real channels only; the pad slot is zero because the buffer is allocated zero-filled, once (advisor finding, round 4)
                dprev = A.get_slack(f"d/bwd/dx{i}/{n}", (n, h, h, ldx), self.gdt, 0) if i == 0 else A.get(f"d/bwd/dx{i}/{n}", (n, h, h, ldx), self.gdt)
                gs = None
                if i > 0 and self.gsum:          # dprev is the gradient at block i-1's InstanceNorm output: its sums come with it
                    gred = A.get(f"d/bwd/gred{i - 1}/{n}", (ops.GSUM_SLOTS * n * cin * 2,), torch.float64)
                    gs = (c["recs"][i - 1]["a"], cin, gred)
                ops.conv2d_dgrad(dz, cout, self.P.op_vars[i], dprev, None, cin, ldx, 0, n, h, h, cin, cout, 3, 2, gsum=gs)
                dcur = dprev
                if i == 4 and params and c["attn"]:          # d attn_disc = sum over the copies of each sample; then its branch
                    B = self._attn_B
                    da = A.get(f"d/bwd/dattn/{B}", (B, h, h, cin), self.gdt)
                    ops.sum_groups(dprev, da, n, h * h * cin, B, 0)
                    self.attn.backward(da)
        return dcur if need_dx else None

    def backward_params(self, drf, dcls):
        """D-loss backward over the whole D batch: fills the flat weight gradient."""
        self._backward(self.ctx["n"], drf, dcls, True, False)
        if self.attn is not None:              # the attention biases were accumulated in f64
            ops.cvt_f64_f32(self.acc, self.P.grad[self.acc_off:], self.P.n - self.acc_off, 0)

    def backward_input(self, n, drf):
        """G-loss backward (data gradient only) through the first n samples."""
        return self._backward(n, drf, None, False, True)

    def backward_input_dz(self, n, drf):
        """Same, but stops at the first layer's pre-activation gradient dz [n,S/2,S/2,F]: the step only needs the
        r+g+b sum of the image gradient, which ops.conv3x3_dgrad_sum1 forms from dz (no 64->3 dgrad)."""
        return self._backward(n, drf, None, False, "dz")

    def attention_masks(self):
        return [(self.attn.ctx[k] > 0).cpu().numpy() for k in ("y1", "y2")]

    def lrelu_tensors(self):
        """The 6 stored LeakyReLU outputs of the last forward (5 convs + patch logits; test diagnostics)."""
        c = self.ctx
        return [r["a"] for r in c["recs"]] + [c["rf"]]

    def lrelu_masks(self):
        """Sign pattern of the 6 LeakyReLU outputs of the last forward (5 convs + patch logits)."""
        return [(t > 0).cpu().numpy() for t in self.lrelu_tensors()]

    def __call__(self, x, training=False, noise=None, keep_mask=None):
        """Keras-style call on [N,S,S,3] (reference: self.D(x, training=...))."""
        n = x.shape[0]
        xd = self.arena.get_slack(f"dcall/x16/{n}", (n, self.S, self.S, self.in_pitch), self.adt, self.pad)
        ops.pack_rgb16(x.contiguous(), noise if training else None, xd, n * self.S * self.S)
        rows = [(0, n, 0)] if (training and keep_mask is not None) else []
        return self.forward(xd, keep_mask, rows)

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        for nme, v in zip(self.names, self.P.vars):
            print_fn(f"{nme:24s} {tuple(v.shape)}  {v.numel()}")
        print_fn(f"Total params: {self.P.n:,}")
