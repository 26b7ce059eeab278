"""PyTorch-CPU (autograd) restatement of SHMGAN's generator, discriminator and train_step.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- also the `cpu_baseline` ("port") leg of
bench.py: the same step with oneDNN convolutions on the host cores.

Follows /root/reference/ShmGANwithSSpecSeg.py ("SHM.py"):
  build_generator      SHM.py:228-327
  build_discriminator  SHM.py:343-389
  train_step           SHM.py:467-875
  gram_matrix          SHM.py:1176-1180
  custom_per_image_standardization  SHM.py:1271-1309
  rescale_01           utils.py:190-195
with the "as executed" facts of SURVEY.md findings 3-7: attention adds a constant zero,
InstanceNormalization has gamma==1 and a constant (untrained) beta, Conv -> bias ->
LeakyReLU(0.2) -> IN order, style factor is an explicit parameter, and the B>1 batch rule
(every sample is an independent B=1 reference step; loss = mean over samples; step-level
RNG draws are shared).

All RNG draws (5 flags, TARGET_LABELS, GaussianNoise, Dropout keep-mask) are explicit inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

LRELU = 0.2
IN_EPS = 1e-6


# ---------------------------------------------------------------------------
# model specification (TF variable order = Keras layer creation order, see
# Generator_summary.txt / Discriminator_summary.txt in the reference)
# ---------------------------------------------------------------------------
def generator_spec(filter_size=64):
    """[(keras_name, kind, k, cin, cout)] in trainable_variables order; every entry owns a
    kernel and a bias.  kind: 'c' = Conv2D s1, 't' = Conv2DTranspose s2."""
    f = filter_size
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


def discriminator_spec(filter_size=64, image_size=128):
    """[(keras_name, kind, shape)] -- no biases anywhere (use_bias=False, SHM.py:366,373,387)."""
    f = filter_size
    s = image_size // 32
    return [
        ("conv2d_27", "c", (3, 3, 3, f)), ("conv2d_28", "c", (3, 3, f, 2 * f)),
        ("conv2d_29", "c", (3, 3, 2 * f, 4 * f)), ("conv2d_30", "c", (3, 3, 4 * f, 8 * f)),
        ("conv2d_33", "c", (3, 3, 8 * f, 16 * f)), ("conv2d_34", "c", (3, 3, 16 * f, 1)),
        ("dense", "d", (s * s * 16 * f, 5)),
    ]


def generator_var_shapes(filter_size=64):
    """Shapes of G.trainable_variables: kernel (HWIO; convT = [kh,kw,Cout,Cin]) then bias."""
    out = []
    for _, kind, k, cin, cout in generator_spec(filter_size):
        out.append((k, k, cin, cout) if kind == "c" else (k, k, cout, cin))
        out.append((cout,))
    return out


def generator_in_channels(filter_size=64):
    """Channel count of each of the 18 InstanceNormalization layers, in layer order."""
    return [cout for name, kind, k, cin, cout in generator_spec(filter_size)
            if kind == "c" and name != "conv2d_26"]


def discriminator_in_channels(filter_size=64):
    f = filter_size
    return [f, 2 * f, 4 * f, 8 * f, 16 * f]


def init_params(filter_size=64, image_size=128, seed=42, beta_seed=43):
    """SURVEY 8(d) synthetic init: weights N(0,0.02) from default_rng(42), biases 0,
    IN beta N(0,0.02) from default_rng(43) (RandomNormal(0,0.02) SHM.py:200)."""
    rng = np.random.default_rng(seed)
    g = []
    for shp in generator_var_shapes(filter_size):
        g.append(np.zeros(shp, np.float32) if len(shp) == 1
                 else rng.normal(0.0, 0.02, shp).astype(np.float32))
    d = [rng.normal(0.0, 0.02, shp).astype(np.float32)
         for _, _, shp in discriminator_spec(filter_size, image_size)]
    brng = np.random.default_rng(beta_seed)
    gb = [brng.normal(0.0, 0.02, (c,)).astype(np.float32) for c in generator_in_channels(filter_size)]
    db = [brng.normal(0.0, 0.02, (c,)).astype(np.float32) for c in discriminator_in_channels(filter_size)]
    return g, d, gb, db


# ---------------------------------------------------------------------------
# "as-intended" live attention branch.  The executed reference evaluates attention_layer on a
# constant zero mask at build time (finding 3); attention="live" feeds the step's SpecSeg mask instead.
# ---------------------------------------------------------------------------
def attention_var_shapes_G(filter_size=64):
    """Variables of the four generator attention_layer calls (SHM.py:248,257,266,275 -> :404-412), per level
    [kernel 1->C, bias, kernel C->C, bias] (Keras Conv2D default use_bias=True), C = F, 2F, 4F, 8F."""
    out = []
    for c in (filter_size, 2 * filter_size, 4 * filter_size, 8 * filter_size):
        out += [(3, 3, 1, c), (c,), (3, 3, c, c), (c,)]
    return out


def attention_var_shapes_D(filter_size=64):
    """attention_layer(filter_size=8F, pool=True, poolsize=(16,16)) of the discriminator (SHM.py:358)."""
    c = 8 * filter_size
    return [(3, 3, 1, c), (c,), (3, 3, c, c), (c,)]


def init_attention(filter_size=64, seed=45, bias_std=0.0):
    """RandomNormal(0, 0.02) kernels (SHM.py:200), zero biases (bias_std > 0: non-trivial biases for tests)."""
    rng = np.random.default_rng(seed)

    def mk(shapes):
        return [(rng.normal(0.0, 0.02, s) if len(s) == 4 else rng.normal(0.0, bias_std, s) if bias_std else np.zeros(s)).astype(np.float32)
                for s in shapes]
    return {"G": mk(attention_var_shapes_G(filter_size)), "D": mk(attention_var_shapes_D(filter_size))}


def attention_layer(av, spec_nchw, pool, masks=None, mi=0):
    """SHM.py:404-412: [MaxPooling2D(pool)] -> Conv2D(3x3, leaky_relu) -> Conv2D(3x3, leaky_relu).
    av = [k1, b1, k2, b2]; returns (attention map, pooled mask)."""
    pooled = F.max_pool2d(spec_nchw, pool) if pool else spec_nchw
    s1 = _act(conv2d_same(pooled, av[0]) + av[1].view(1, -1, 1, 1), masks, mi)
    s2 = _act(conv2d_same(s1, av[2]) + av[3].view(1, -1, 1, 1), masks, mi + 1)
    return s2, pooled


# ---------------------------------------------------------------------------
# TF op semantics in torch (NHWC at the interface, NCHW inside)
# ---------------------------------------------------------------------------
def _same_pads(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv2d_same(x, w_hwio, stride=1):
    """x NCHW, w HWIO.  TF SAME (asymmetric for stride 2 on even sizes: 0 before, 1 after)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    pt, pb = _same_pads(x.shape[2], kh, stride)
    pl, pr = _same_pads(x.shape[3], kw, stride)
    if pt or pb or pl or pr:
        x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1).contiguous(), stride=stride)


def conv2d_transpose_same(x, w_hwoi, stride=2):
    """Keras Conv2DTranspose 'same': w is [kh,kw,Cout,Cin]; output = in*stride; defined as
    the input-gradient of the SAME stride-s conv, whose pad_before is 0 for k=3,s=2."""
    kh, kw = w_hwoi.shape[0], w_hwoi.shape[1]
    H, W = x.shape[2] * stride, x.shape[3] * stride
    pt, _ = _same_pads(H, kh, stride)
    pl, _ = _same_pads(W, kw, stride)
    y = F.conv_transpose2d(x, w_hwoi.permute(3, 2, 0, 1).contiguous(), stride=stride)
    return y[:, :, pt:pt + H, pl:pl + W]


def instance_norm(x, beta):
    """tfa InstanceNormalization op chain (Generator_summary.txt:9-36), gamma == 1."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mean.detach()) ** 2).mean(dim=(2, 3), keepdim=True)
    inv = torch.rsqrt(var + IN_EPS)
    return x * inv + (beta.view(1, -1, 1, 1) - mean * inv)


class KinkRecorder:
    """Collects, per (pass tag, layer), the pre-activations the float64 step finds within `thr` of LeakyReLU's kink,
    as flat NHWC indices into the DEVICE's batched tensors ([B] rows for "g1", [5B] k-major for "cyc", [12B] ordered
    [D1][D3 x5][D2][D4 x5] for "d") together with the side of zero the float64 value lies on.  A float32 run can only
    disagree with float64 about the sign of such elements; tests/test_step_gpu.py pins the device to the recorded side at
    the (few) elements where it does, which makes an un-pinned fixture comparable to 1e-3 per tensor (the reverse of the
    `masks=` pinning of _act).  batch_total / sample_offset: the fixture is generated one sample at a time."""

    def __init__(self, thr=1e-4, batch_total=1, sample_offset=0):
        self.thr, self.B, self.b0 = float(thr), int(batch_total), int(sample_offset)
        self.tag, self.row0 = None, 0
        self.items = {}

    def at(self, tag, group):
        """Next forward call is device pass `tag`, rows starting at group * batch_total + sample_offset."""
        self.tag, self.row0 = tag, group * self.B + self.b0

    def add(self, layer, z):
        zd = z.detach()
        zn = zd.permute(0, 2, 3, 1) if zd.dim() == 4 else zd          # NCHW -> NHWC
        near = (zn.abs() < self.thr).reshape(zn.shape[0], -1)
        r, off = torch.nonzero(near, as_tuple=True)
        per = near.shape[1]
        idx = (r + self.row0) * per + off
        pos = zn.reshape(zn.shape[0], -1)[r, off] > 0
        key = (self.tag, layer)
        old = self.items.get(key)
        self.items[key] = (idx, pos) if old is None else (torch.cat([old[0], idx]), torch.cat([old[1], pos]))

    def to_npz(self):
        out = {"kink/thr": np.float64(self.thr)}
        for (tag, layer), (idx, pos) in sorted(self.items.items()):
            o = torch.argsort(idx)
            out[f"kink/{tag}/{layer:02d}/idx"] = idx[o].numpy().astype(np.int64)
            out[f"kink/{tag}/{layer:02d}/pos"] = pos[o].numpy().astype(np.bool_)
        return out


_KINKS = None          # active KinkRecorder (set by train_step(kinks=...) for the duration of the call)


def _act(z, masks, idx):
    """LeakyReLU(0.2).  `masks` (optional list of NHWC bool arrays, one per activation in layer
    order) pins which side of the kink every element is evaluated on: a float32 device can round a
    |z| < 1e-6 pre-activation to the other side of 0, where the derivative jumps 0.2 <-> 1; giving
    the oracle the device's sign pattern removes those (legitimate) discontinuity events from a
    gradient comparison.  The forward value changes by at most 0.8*|z| ~ 1e-6 there."""
    if _KINKS is not None:
        _KINKS.add(idx, z)
    if masks is None:
        return F.leaky_relu(z, LRELU)
    m = torch.as_tensor(np.asarray(masks[idx])).bool()
    m = m.permute(0, 3, 1, 2) if m.dim() == 4 else m
    return torch.where(m, z, LRELU * z)


def generator_attention(avars, mask_nhwc, masks=None):
    """The four attention maps attn_1..4 of build_generator (SHM.py:248,257,266,275): the first on the mask itself
    (pool=False), each next one on the 2x2 max-pool of the previous level's pooled mask."""
    spec = mask_nhwc.permute(0, 3, 1, 2)
    out = []
    for lvl in range(4):
        a, spec = attention_layer(avars[4 * lvl:4 * lvl + 4], spec, 2 if lvl else None, masks, 2 * lvl)
        out.append(a)
    return out


def generator_forward(gvars, gbetas, x_nhwc, filter_size=64, record=None, masks=None, attn=None):
    """SHM.py:228-327.  x [B,S,S,10] -> [B,S,S,1].  record: optional list that receives, per
    Conv->LReLU->IN block, (pre-activation z, IN output) with retain_grad (test diagnostics).
    attn: optional [attn_1..attn_4] (NCHW, batch B or a divisor of it: tiled) added to the skips (SHM.py:290-293);
    None = the executed graph (adds zeros)."""
    spec = generator_spec(filter_size)
    x = x_nhwc.permute(0, 3, 1, 2)
    vi = [0]
    bi = [0]

    def nxt():
        w, b = gvars[vi[0]], gvars[vi[0] + 1]
        vi[0] += 2
        return w, b

    def cnl(x):           # Conv2D(act=leaky_relu) -> InstanceNormalization
        w, b = nxt()
        z = conv2d_same(x, w) + b.view(1, -1, 1, 1)
        x = instance_norm(_act(z, masks, vi[0] // 2 - 1), gbetas[bi[0]])
        if record is not None:
            if z.requires_grad:
                z.retain_grad()
                x.retain_grad()
            record.append((z, x))
        bi[0] += 1
        return x

    downs = []
    for lvl in range(4):
        x = cnl(cnl(x))
        downs.append(x)                       # + attn_k == + 0 (finding 3)
        x = F.avg_pool2d(x, 2)
    x = cnl(cnl(x))                           # the two 1x1 layers (SHM.py:280-282)
    if attn is not None:                      # down_k = down_k + attn_k (SHM.py:290-293): only the skip, not the pooled path
        downs = [d + a.repeat(d.shape[0] // a.shape[0], 1, 1, 1) for d, a in zip(downs, attn)]
    for lvl in range(4):
        w, b = nxt()
        x = _act(conv2d_transpose_same(x, w) + b.view(1, -1, 1, 1), masks, vi[0] // 2 - 1)
        x = torch.cat([x, downs[3 - lvl]], dim=1)       # upsampled first, skip second
        x = cnl(cnl(x))
    w, b = nxt()
    x = _act(conv2d_same(x, w) + b.view(1, -1, 1, 1), masks, vi[0] // 2 - 1)
    assert vi[0] == len(spec) * 2 and bi[0] == len(gbetas)
    return x.permute(0, 2, 3, 1)


def discriminator_forward(dvars, dbetas, x_nhwc, noise=None, keep_mask=None, dropout=0.2, masks=None, attn=None):
    """SHM.py:343-389.  x [B,S,S,3] -> ([B,s,s,1], [B,5]).  training=True <=> noise and
    keep_mask given: GaussianNoise adds `noise`; Dropout multiplies by keep_mask/(1-rate).
    attn: optional attn_disc (NCHW, [B, 8F, S/16, S/16]) added after the fourth block (SHM.py:358-359)."""
    x = x_nhwc
    if noise is not None:
        x = x + noise
    x = x.permute(0, 3, 1, 2)
    for i in range(5):
        x = _act(conv2d_same(x, dvars[i], stride=2), masks, i)
        x = instance_norm(x, dbetas[i])
        if i == 3 and attn is not None:       # x = x + attn_disc (SHM.py:359); the executed graph adds zeros
            x = x + attn
    if keep_mask is not None:
        x = x * keep_mask.permute(0, 3, 1, 2) / (1.0 - dropout)
    rf = _act(conv2d_same(x, dvars[5]), masks, 5).permute(0, 2, 3, 1)
    flat = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)       # Flatten of NHWC
    cls = flat @ dvars[6]
    return rf, cls


_RGB2YUV = [[0.299, -0.14714119, 0.61497538],
            [0.587, -0.28886916, -0.51496512],
            [0.114, 0.43601035, -0.10001026]]
_YUV2RGB = [[1.0, 1.0, 1.0],
            [0.0, -0.394642334, 2.03206185],
            [1.13988303, -0.58062185, 0.0]]


def rgb_to_yuv(x):
    return x @ torch.tensor(_RGB2YUV, dtype=x.dtype)


def yuv_to_rgb(x):
    return x @ torch.tensor(_YUV2RGB, dtype=x.dtype)


def per_image_standardization(x):
    """SHM.py:1271-1309, per sample.  Returns (x/scale, scale[B])."""
    m = x.mean(dim=(1, 2, 3))
    var = torch.relu((x * x).mean(dim=(1, 2, 3)) - m * m)
    scale = torch.maximum(torch.sqrt(var), torch.tensor(1.0 / 256.0, dtype=x.dtype))
    return x / scale.view(-1, 1, 1, 1), scale


def rescale_01(x):
    """utils.py:190-195 per sample; divide_no_nan; gradient flows through min and max."""
    mn = x.amin(dim=(1, 2, 3), keepdim=True)
    mx = x.amax(dim=(1, 2, 3), keepdim=True)
    den = mx - mn
    safe = torch.where(den == 0, torch.ones_like(den), den)
    return torch.where(den == 0, torch.zeros_like(x), (x - mn) / safe)


def _gauss_window(dtype, size=11, sigma=1.5):
    c = torch.arange(size, dtype=torch.float64) - (size - 1) / 2.0
    g = -0.5 * c * c / (sigma * sigma)
    g2 = (g[None, :] + g[:, None]).reshape(-1)
    return torch.softmax(g2, 0).reshape(size, size).to(dtype)


def ssim(x, y, max_val=5.0, k1=0.01, k2=0.03):
    """tf.image.ssim(x, y, 5) (SHM.py:759-763) -> [B]."""
    c = x.shape[-1]
    win = _gauss_window(x.dtype).view(1, 1, 11, 11).repeat(c, 1, 1, 1)
    xn, yn = x.permute(0, 3, 1, 2), y.permute(0, 3, 1, 2)

    def filt(z):
        return F.conv2d(z, win, groups=c)

    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mx, my = filt(xn), filt(yn)
    num0 = mx * my * 2.0
    den0 = mx * mx + my * my
    lum = (num0 + c1) / (den0 + c1)
    num1 = filt(xn * yn) * 2.0
    den1 = filt(xn * xn + yn * yn)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(dim=(2, 3)).mean(dim=1)


def gram_matrix(x):
    return torch.einsum('bijc,bijd->bcd', x, x) / float(x.shape[1] * x.shape[2])


def style_factor_intended(image_size):
    """As-intended 1/(2*9*S*S)^2 (SHM.py:817 overflows int32 at S>=256: SURVEY finding 7)."""
    return 1.0 / float(2 * 9 * image_size * image_size) ** 2


def style_factor_as_executed(image_size):
    """tf.math.square on a Python int is an int32 op: the square wraps mod 2^32 (finding 7)."""
    v = (2 * 9 * image_size * image_size) ** 2
    v32 = ((v + 2 ** 31) % 2 ** 32) - 2 ** 31
    return float('inf') if v32 == 0 else 1.0 / v32


# ---------------------------------------------------------------------------
# step-level RNG draws (explicit inputs)
# ---------------------------------------------------------------------------
@dataclass
class StepDraws:
    flags: tuple            # 5 bools: RNG1..RNG5 (True -> view zeroed / replaced by gen_Y)  SHM.py:509-513
    target_label: float     # self.TARGET_LABELS, U(0.8,1.2) per step (SHM.py:986)
    noise: np.ndarray       # [2B,S,S,3]: GaussianNoise(0.1) for D1 (first B) then D2  (SHM.py:352,559,563)
    keep_mask: np.ndarray   # [2B,s,s,16F] in {0,1}: Dropout(0.2) keep mask, D1 then D2   (SHM.py:363)


def make_draws(step, batch, image_size, filter_size, rank=0):
    """flags/TARGET_LABELS from default_rng(7+step) (shared across ranks);
    noise/dropout from default_rng([7+step, rank])."""
    r = np.random.default_rng(7 + step)
    flags = tuple(bool(u < 0.5) for u in r.random(5))
    target = float(r.uniform(0.8, 1.2))
    r2 = np.random.default_rng([7 + step, rank, 1])
    s = image_size // 32
    noise = (0.1 * r2.standard_normal((2 * batch, image_size, image_size, 3))).astype(np.float32)
    keep = (r2.random((2 * batch, s, s, 16 * filter_size)) >= 0.2).astype(np.float32)
    return StepDraws(flags, target, noise, keep)


def make_inputs(batch, image_size, rank=0):
    """Five [B,S,S,3] fp32 U[0,1) tensors from default_rng(1234+rank)."""
    r = np.random.default_rng(1234 + rank)
    return [r.random((batch, image_size, image_size, 3), dtype=np.float32) for _ in range(5)]


@dataclass
class AdamState:
    m: list
    v: list
    iterations: int = 0


def adam_apply(params, grads, st: AdamState, lr0, beta1, beta2, eps=1e-7):
    """clip_by_value(+-1) then Keras adam_v2 with ExponentialDecay(lr0,10000,0.95)
    (SHM.py:169-175, 859-872).  In place on `params` (list of tensors)."""
    t = st.iterations + 1
    lr = lr0 * 0.95 ** (st.iterations / 10000.0)
    alpha = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, st.m, st.v):
            g = g.clamp(-1.0, 1.0)
            m += (g - m) * (1.0 - beta1)
            v += (g * g - v) * (1.0 - beta2)
            p -= alpha * m / (v.sqrt() + eps)
    st.iterations += 1


# ---------------------------------------------------------------------------
# the step
# ---------------------------------------------------------------------------
class _XentTFFused(torch.autograd.Function):
    """tf.nn.softmax_cross_entropy_with_logits(labels, logits) AS EXECUTED by TensorFlow 2.8 (call sites SHM.py:695-713).
    Forward: -sum(labels * log_softmax(logits)) per row.  Backward wrt logits: the op's second output
    `backprop = softmax(logits) - labels` (tensorflow/core/kernels/xent_op.h) times the incoming gradient
    (`_SoftmaxCrossEntropyWithLogitsGrad`, tensorflow/python/ops/nn_grad.py: `grad = _BroadcastMul(grad_loss,
    op.outputs[1])`).  That is the derivative only when every label row sums to 1; the D1 term's row is
    [0,0,0,0,TARGET_LABELS] (SHM.py:477, 533, 688, 702) with TARGET_LABELS ~ U(0.8, 1.2) (SHM.py:986), for which the
    true derivative would be T*softmax - labels.  Labels are constants here (no gradient wrt labels is needed).
    TensorFlow is not installable in this container, so this is restated from TF's published sources, not run against it."""

    @staticmethod
    def forward(ctx, logits, labels):
        logp = torch.log_softmax(logits, dim=-1)
        ctx.save_for_backward(logp.exp() - labels)
        return -(labels * logp).sum(dim=-1)

    @staticmethod
    def backward(ctx, grad_loss):
        (backprop,) = ctx.saved_tensors
        return grad_loss.unsqueeze(-1) * backprop, None


def softmax_xent(logits, labels, mode="executed"):
    """Per-row softmax cross entropy [B]: mode "executed" = TF's fused kernel and its registered gradient (_XentTFFused),
    "intended" = plain autograd of -sum(labels * log_softmax) (the true derivative).  Values are identical."""
    assert mode in ("executed", "intended")
    if mode == "executed":
        return _XentTFFused.apply(logits, labels)
    return -(labels * torch.log_softmax(logits, dim=-1)).sum(dim=-1)


def _mslice(masks, lo, hi):
    return None if masks is None else [np.asarray(m)[lo:hi] for m in masks]


def train_step(gvars, dvars, gbetas, dbetas, inputs, draws: StepDraws, style_factor,
               filter_size=64, dtype=torch.float64, need_grads=True, masks=None, specseg=None, attention=None,
               xent_mode="executed", kinks=None):
    """One SHM.py:467-875 forward + both tape.gradient calls (no optimizer apply).

    gvars/dvars/gbetas/dbetas: lists of numpy arrays or tensors (TF layouts).
    inputs: 5 x [B,S,S,3] in [0,1].  Returns dict(losses=..., gG=[...], gD=[...], outs=...).
    masks: optional {"g1": [23 x [B,...]], "cyc": [23 x [5B,...]], "d": [6 x [12B,...]]} LeakyReLU
    sign patterns taken from the device run (see _act); D batch order [D1][D3 x5][D2][D4 x5].
    specseg: optional SpecSeg weights (oracle.specseg_torch layout): adds the mask of SHM.py:492 to
    outs["specular_candidate"] and the logged-only losses["Spec_loss"] (SHM.py:792-806).
    attention: optional {"G": [16 arrays], "D": [4 arrays]} (init_attention): the "as-intended" LIVE attention branch --
    the step's SpecSeg mask (needs `specseg`; a constant, no gradient flows into SpecSeg) through attention_layer into
    the four generator skips and the discriminator, for every G / D call of the step.  The result then carries gGa / gDa,
    the gradients of the attention variables (G loss / D loss).  masks may hold "ga" (8) and "da" (2) sign patterns.
    xent_mode: "executed" (default) = the class-logit gradient of TF's fused softmax-cross-entropy kernel
    (softmax - labels, see _XentTFFused); "intended" = the true derivative.  Only D1's label row [0,0,0,0,T] tells them apart.
    kinks: optional KinkRecorder that receives the near-kink pre-activations of every LeakyReLU of the step.
    """
    global _KINKS
    _KINKS = kinks
    try:
        return _train_step(gvars, dvars, gbetas, dbetas, inputs, draws, style_factor, filter_size, dtype, need_grads, masks,
                           specseg, attention, xent_mode, kinks)
    finally:
        _KINKS = None


def _train_step(gvars, dvars, gbetas, dbetas, inputs, draws, style_factor, filter_size, dtype, need_grads, masks, specseg,
                attention, xent_mode, kinks):
    at = kinks.at if kinks is not None else (lambda tag, group: None)
    mk = masks or {}
    T = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dtype)
    gv = [T(a).clone().requires_grad_(need_grads) for a in gvars]
    dv = [T(a).clone().requires_grad_(need_grads) for a in dvars]
    gb = [T(a) for a in gbetas]
    db = [T(a) for a in dbetas]
    orig = [T(a) for a in inputs]
    B, S = orig[0].shape[0], orig[0].shape[1]
    tl = float(draws.target_label)
    flags = [bool(f) for f in draws.flags]
    noise = T(draws.noise)
    keep = T(draws.keep_mask)

    # pre-processing (outside the tape)  SHM.py:480-490
    ds, scales = [], []
    for o in orig:
        y, sc = per_image_standardization(rgb_to_yuv(o))
        ds.append(y)
        scales.append(sc)
    Ych = [d[..., 0:1] for d in ds]
    zeros = torch.zeros(B, S, S, 1, dtype=dtype)
    ones = torch.ones(B, S, S, 1, dtype=dtype)
    avgCbCr = (ds[0][..., 1:] + ds[1][..., 1:] + ds[2][..., 1:] + ds[3][..., 1:] + ds[4][..., 1:]) / 5.0

    # specular mask (outside the tape, SHM.py:492) and, in live mode, the attention maps it produces
    spec_mask = None
    if specseg is not None:
        from .specseg_torch import specseg_forward
        spec_mask = specseg_forward(specseg, Ych[2], dtype).detach()
    ga = da = attn_g = attn_d = None
    if attention is not None:
        assert spec_mask is not None, "attention='live' needs the SpecSeg weights"
        ga = [T(a).clone().requires_grad_(need_grads) for a in attention["G"]]
        da = [T(a).clone().requires_grad_(need_grads) for a in attention["D"]]
        at("ga", 0)
        attn_g = generator_attention(ga, spec_mask, mk.get("ga"))
        at("da", 0)
        attn_d, _ = attention_layer(da, spec_mask.permute(0, 3, 1, 2), 16, mk.get("da"), 0)

    # G(1)  SHM.py:517-538
    rand_inp = [zeros if flags[k] else Ych[k] for k in range(5)]
    gen_input = torch.cat(rand_inp + [zeros, zeros, zeros, zeros, ones], dim=3)
    at("g1", 0)
    gen_Y = generator_forward(gv, gb, gen_input, filter_size, masks=mk.get("g1"), attn=attn_g)
    gen_yuv = torch.cat([gen_Y, avgCbCr], dim=3)
    gen_rgb = yuv_to_rgb(gen_yuv)

    # D(1), D(2): training=True  SHM.py:559-563
    md = mk.get("d")
    at("d", 0)
    rf_D1, cls_D1 = discriminator_forward(dv, db, gen_rgb, noise[:B], keep[:B], masks=_mslice(md, 0, B), attn=attn_d)
    at("d", 6)
    rf_D2, cls_D2 = discriminator_forward(dv, db, orig[4], noise[B:], keep[B:], masks=_mslice(md, 6 * B, 7 * B), attn=attn_d)

    # G(2): cyclic inputs  SHM.py:576-607
    sub = [gen_Y if flags[k] else Ych[k] for k in range(5)]
    cyc_Y = []
    for k in range(5):
        chans = [zeros if j == k else sub[j] for j in range(5)]
        onehot = [ones if j == k else zeros for j in range(5)]
        at("cyc", k)
        cyc_Y.append(generator_forward(gv, gb, torch.cat(chans + onehot, dim=3), filter_size,
                                       masks=_mslice(mk.get("cyc"), k * B, (k + 1) * B), attn=attn_g))
    cyc_yuv = [torch.cat([cy, avgCbCr], dim=3) for cy in cyc_Y]
    cyc_rgb = [yuv_to_rgb(c) for c in cyc_yuv]

    # D(3), D(4): training=False  SHM.py:627-642
    D3, D4 = [], []
    for k in range(5):
        at("d", 1 + k)
        D3.append(discriminator_forward(dv, db, cyc_rgb[k], masks=_mslice(md, (1 + k) * B, (2 + k) * B), attn=attn_d))
    for k in range(5):
        at("d", 7 + k)
        D4.append(discriminator_forward(dv, db, orig[k], masks=_mslice(md, (7 + k) * B, (8 + k) * B), attn=attn_d))

    def mse(a, t):      # per-sample mean -> [B]
        return ((a - t) ** 2).mean(dim=(1, 2, 3))

    def xent(logits, k, w=1.0):          # label row = w * onehot(k)
        lab = torch.zeros_like(logits)
        lab[:, k] = w
        return softmax_xent(logits, lab, xent_mode)

    # losses SHM.py:669-844 (per-sample [B] vectors; reduced with mean at the end)
    D3_RF = sum(mse(D3[k][0], tl) for k in range(5))
    D1_RF = mse(rf_D1, tl)
    D3_cls = sum(xent(D3[k][1], k) for k in range(5))
    D1_cls = xent(cls_D1, 4, tl)                          # labels [0,0,0,0,T]  SHM.py:477,688,702
    D4_cls = sum(xent(D4[k][1], k) for k in range(5))
    D2_RF = mse(rf_D2, tl) + (rf_D1 ** 2).mean(dim=(1, 2, 3))
    D4_RF = sum(mse(D4[k][0], tl) + (D3[k][0] ** 2).mean(dim=(1, 2, 3)) for k in range(5)) + D2_RF

    l1 = lambda a, b: (a - b).abs().mean(dim=(1, 2, 3))
    L1_G1 = l1(gen_rgb, orig[4])
    L1_c = [l1(cyc_rgb[k], orig[k]) for k in range(5)]
    L1_loss = (L1_c[0] + L1_c[1] + L1_c[2] + L1_c[3] + L1_G1) / 5.0 + L1_c[4] * 10.0

    ssims = [ssim(rescale_01(cyc_yuv[k]), rescale_01(ds[k])) for k in range(5)]
    sl = [torch.zeros(B, dtype=dtype) if flags[k] else -torch.log((1.0 + ssims[k]) / 2.0) for k in range(5)]
    ssim_loss = (sl[0] + sl[1] + sl[2] + sl[3] + sl[4] * 10.0) / 5.0

    content = ((cyc_yuv[4] - ds[0]) ** 2).mean(dim=(1, 2, 3))
    style = style_factor * ((gram_matrix(cyc_yuv[4]) - gram_matrix(ds[4])) ** 2).mean(dim=(1, 2))
    nst = 100.0 * style + content

    total_G = (D1_RF + D3_RF) / 6.0 + 10.0 * L1_loss + 10.0 * ssim_loss + 10.0 * nst
    total_D = (D1_cls + D3_cls) / 6.0 + (D2_RF + D4_RF) / 6.0 + 0.5 * D4_cls + 10.0 * nst
    total_C = (D4_cls + nst) * 10.0

    losses = {
        "total_Generator_loss": total_G, "total_Discriminator_loss": total_D,
        "total_Classification_loss": total_C, "G_gan_loss": (D3_RF + D1_RF) / 6.0,
        "G_clsf_loss": (D3_cls + D1_cls) / 6.0,
        "D1_RealFake_loss": D1_RF, "D3_RealFake_cyc": D3_RF, "D2_RealFake_target": D2_RF,
        "D4_RealFake_cyc": D4_RF, "D1_classification_loss": D1_cls,
        "D3_classification_loss": D3_cls, "D4_classification_loss": D4_cls,
        "L1_loss_Gen": L1_loss, "ssim_cyc_loss": ssim_loss, "content_loss": content,
        "style_loss": style, "total_NST_loss": nst,
    }
    if specseg is not None:
        from .specseg_torch import spec_loss
        losses["Spec_loss"] = spec_loss([c.detach() for c in cyc_yuv], ds, spec_mask)[0]      # mask = predict(I90_Ych)  SHM.py:492
    out = {"losses": {k: float(v.mean().detach()) for k, v in losses.items()},
           "outs": {"gen_Y": gen_Y.detach(), "gen_rgb": gen_rgb.detach(),
                    "cyc_rgb": [c.detach() for c in cyc_rgb],
                    "rf_D1": rf_D1.detach(), "cls_D1": cls_D1.detach(),
                    "ssim": [s_.detach() for s_ in ssims], "scales": [s_.detach() for s_ in scales],
                    "specular_candidate": spec_mask}}
    if need_grads:
        gD = torch.autograd.grad((total_D + total_C).mean(), dv + (da or []), retain_graph=True)
        if da is not None:
            gD, out["gDa"] = gD[:len(dv)], list(gD[len(dv):])
        # intermediate generator-loss gradients (test diagnostics): wrt the 5 cyclic outputs and
        # the total derivative wrt gen_Y (direct terms + the G o G chain through the cyclic inputs)
        mid = torch.autograd.grad(total_G.mean(), cyc_Y + [gen_Y], retain_graph=True)
        out["dcyc_Y"], out["dgen_Y"] = [m_.detach() for m_ in mid[:5]], mid[5].detach()
        gG = torch.autograd.grad(total_G.mean(), gv + (ga or []))
        if ga is not None:
            gG, out["gGa"] = gG[:len(gv)], list(gG[len(gv):])
        out["gD"], out["gG"] = list(gD), list(gG)
    return out


# ---------------------------------------------------------------------------
# inference path of the evaluation script
# ---------------------------------------------------------------------------
def infer(gvars, gbetas, rgb, filter_size=64, dtype=torch.float64):
    """/root/reference/test.py:218-297 restated: one RGB image in, the generated (ED) image and the
    five cyclic reconstructions out.  rgb [B,S,S,3] in [0,1].  Returns dict of NHWC tensors."""
    T = lambda a: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(dtype)
    gv = [T(a) for a in gvars]
    gb = [T(a) for a in gbetas]
    x = T(rgb)
    B, S = x.shape[0], x.shape[1]
    yuv, scale = per_image_standardization(rgb_to_yuv(x))                 # test.py:218
    cbcr = yuv[..., 1:]                                                   # test.py:224
    zeros = torch.zeros(B, S, S, 1, dtype=dtype)
    ones = torch.ones(B, S, S, 1, dtype=dtype)
    gen_input = torch.cat([yuv[..., 0:1]] + [zeros] * 8 + [ones], dim=3)  # test.py:227-237
    gen_Y = generator_forward(gv, gb, gen_input, filter_size)             # test.py:243
    gen_rgb = yuv_to_rgb(torch.cat([gen_Y, cbcr], dim=3))                 # test.py:244-250
    orig_Ych = gen_rgb[..., 0:1]                                          # test.py:252 (channel 0 of the RGB)
    cyc_rgb = []
    for k in range(5):                                                    # test.py:260-297
        chans = [zeros if j == k else orig_Ych for j in range(5)]
        onehot = [ones if j == k else zeros for j in range(5)]
        cy = generator_forward(gv, gb, torch.cat(chans + onehot, dim=3), filter_size)
        cyc_rgb.append(yuv_to_rgb(torch.cat([cy, cbcr], dim=3)))
    return {"gen_Y": gen_Y, "gen_rgb": gen_rgb, "cyc_rgb": cyc_rgb, "scale": scale}
