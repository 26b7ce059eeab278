"""NumPy restatement of the TensorFlow/Keras op semantics the SHMGAN hot path uses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Explicit index arithmetic, no
torch, float64 by default.  Each function cites the reference call site
(``SHM.py`` = /root/reference/ShmGANwithSSpecSeg.py) whose TF op it restates.
These are the "slow but obviously right" forms used to pin the torch oracle
(``step_torch``) and, at tiny sizes, the HIP kernels.
"""
from __future__ import annotations

import numpy as np

LRELU_ALPHA = 0.2          # tf.nn.leaky_relu default alpha (SHM.py:244 activation=tf.nn.leaky_relu)
IN_EPS = 1e-6              # InstanceNormalization(epsilon=0.000001) SHM.py:245


# ---------------------------------------------------------------------------
# TF "SAME" padding rule (tensorflow/core/framework/common_shape_fns.cc,
# GetWindowedOutputSizeVerbose): out = ceil(in/s); pad_total = max((out-1)*s+k-in, 0);
# pad_before = pad_total // 2; the remainder goes after.
# ---------------------------------------------------------------------------
def same_pads(in_size: int, k: int, s: int):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def leaky_relu(x, alpha=LRELU_ALPHA):
    return np.where(x > 0, x, alpha * x)


def leaky_relu_grad(x, dy, alpha=LRELU_ALPHA):
    # TF LeakyReluGrad: features > 0 ? g : alpha * g
    return np.where(x > 0, dy, alpha * dy)


def conv2d_same(x, w, stride=1):
    """Keras Conv2D(padding='same') core (SHM.py:244, :387, :365). x NHWC, w HWIO."""
    n, h, wd, ci = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2
    ho, pt, pb = same_pads(h, kh, stride)
    wo, pl, pr = same_pads(wd, kw, stride)
    xp = np.zeros((n, h + pt + pb, wd + pl + pr, ci), x.dtype)
    xp[:, pt:pt + h, pl:pl + wd] = x
    y = np.zeros((n, ho, wo, co), x.dtype)
    for a in range(kh):
        for b in range(kw):
            patch = xp[:, a:a + (ho - 1) * stride + 1:stride, b:b + (wo - 1) * stride + 1:stride]
            y += patch @ w[a, b]
    return y


def conv2d_same_bwd(x, w, dy, stride=1):
    """Gradients of conv2d_same wrt x and w."""
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    ho, pt, pb = same_pads(h, kh, stride)
    wo, pl, pr = same_pads(wd, kw, stride)
    xp = np.zeros((n, h + pt + pb, wd + pl + pr, ci), x.dtype)
    xp[:, pt:pt + h, pl:pl + wd] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for a in range(kh):
        for b in range(kw):
            sl = (slice(None), slice(a, a + (ho - 1) * stride + 1, stride),
                  slice(b, b + (wo - 1) * stride + 1, stride))
            dw[a, b] = np.einsum('nhwc,nhwd->cd', xp[sl], dy)
            dxp[sl] += dy @ w[a, b].T
    return dxp[:, pt:pt + h, pl:pl + wd], dw


def conv2d_transpose_same(x, w, stride=2):
    """Keras Conv2DTranspose(padding='same') core (SHM.py:298,305,312,319).

    x NHWC [n,h,w,ci]; w is [kh,kw,Cout,Cin] (Keras layout).  Output is
    [n, h*stride, w*stride, Cout].  tf.nn.conv2d_transpose is *defined* as the
    input-gradient of the SAME conv2d that maps the output back to x, so the
    scatter below uses that forward conv's pad_before.
    """
    n, h, wd, ci = x.shape
    kh, kw, co, ci2 = w.shape
    assert ci == ci2
    H, W = h * stride, wd * stride
    _, pt, _ = same_pads(H, kh, stride)
    _, pl, _ = same_pads(W, kw, stride)
    y = np.zeros((n, H, W, co), x.dtype)
    for i in range(h):
        for a in range(kh):
            p = i * stride + a - pt
            if not 0 <= p < H:
                continue
            for j in range(wd):
                for b in range(kw):
                    q = j * stride + b - pl
                    if 0 <= q < W:
                        y[:, p, q] += x[:, i, j] @ w[a, b].T
    return y


def instance_norm(x, beta, gamma=None, eps=IN_EPS):
    """tfa InstanceNormalization as lowered in Generator_summary.txt:9-36:
    mean -> squared_difference -> mean -> +eps -> rsqrt -> *gamma -> x*inv + (beta - mean*inv)."""
    mean = x.mean(axis=(1, 2), keepdims=True)
    var = ((x - mean) ** 2).mean(axis=(1, 2), keepdims=True)
    inv = 1.0 / np.sqrt(var + eps)
    if gamma is not None:
        inv = inv * gamma
    return x * inv + (beta - mean * inv)


def instance_norm_bwd(x, dy, eps=IN_EPS):
    """dx for gamma == 1 (SURVEY finding 4: gamma/beta are untracked constants)."""
    mean = x.mean(axis=(1, 2), keepdims=True)
    var = ((x - mean) ** 2).mean(axis=(1, 2), keepdims=True)
    inv = 1.0 / np.sqrt(var + eps)
    xh = (x - mean) * inv
    return inv * (dy - dy.mean(axis=(1, 2), keepdims=True) - xh * (dy * xh).mean(axis=(1, 2), keepdims=True))


def avg_pool2(x):
    """AveragePooling2D(2,2,'same') on even sizes (SHM.py:249)."""
    n, h, w, c = x.shape
    assert h % 2 == 0 and w % 2 == 0
    return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4))


# tf.image.rgb_to_yuv / yuv_to_rgb kernels (tensorflow/python/ops/image_ops_impl.py),
# applied as tensordot(images, kernel, axes=[[-1],[0]]) -- SHM.py:480-484, :553, :620-624.
RGB2YUV = np.array([[0.299, -0.14714119, 0.61497538],
                    [0.587, -0.28886916, -0.51496512],
                    [0.114, 0.43601035, -0.10001026]])
YUV2RGB = np.array([[1.0, 1.0, 1.0],
                    [0.0, -0.394642334, 2.03206185],
                    [1.13988303, -0.58062185, 0.0]])


def rgb_to_yuv(x):
    return x @ RGB2YUV.astype(x.dtype)


def yuv_to_rgb(x):
    return x @ YUV2RGB.astype(x.dtype)


def per_image_standardization(x):
    """custom_per_image_standardization (SHM.py:1271-1309): x / max(std, 1/256), NO mean
    subtraction (:1301 commented out); statistics over the whole per-sample tensor."""
    out = np.empty_like(x)
    scales = []
    for i in range(x.shape[0]):
        m = x[i].mean()
        var = max((x[i] ** 2).mean() - m * m, 0.0)
        scale = max(np.sqrt(var), 1.0 / 256.0)      # rsqrt(65536) SHM.py:1280,1293
        out[i] = x[i] / scale
        scales.append(scale)
    return out, np.array(scales)


def rescale_01(x):
    """utils.py:190-195, per sample (batch rule); divide_no_nan."""
    out = np.empty_like(x)
    for i in range(x.shape[0]):
        mn, mx = x[i].min(), x[i].max()
        out[i] = 0.0 if mx == mn else (x[i] - mn) / (mx - mn)
    return out


def gauss_window(size=11, sigma=1.5, dtype=np.float64):
    """tf.image ssim _fspecial_gauss: softmax over the 2-D grid of -(x^2+y^2)/(2 sigma^2)."""
    c = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = -0.5 * c * c / (sigma * sigma)
    g2 = g[None, :] + g[:, None]
    e = np.exp(g2 - g2.max())
    return (e / e.sum()).astype(dtype)


def ssim(x, y, max_val=5.0, k1=0.01, k2=0.03, size=11, sigma=1.5):
    """tf.image.ssim(x, y, max_val) (SHM.py:759-763) -> [B].  Per channel, VALID depthwise
    gaussian; ssim = mean_{h,w}(luminance*cs); mean over channels."""
    n, h, w, c = x.shape
    win = gauss_window(size, sigma, x.dtype)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    ho, wo = h - size + 1, w - size + 1

    def filt(z):
        out = np.zeros((n, ho, wo, c), z.dtype)
        for a in range(size):
            for b in range(size):
                out += win[a, b] * z[:, a:a + ho, b:b + wo]
        return out

    mx, my = filt(x), filt(y)
    num0 = mx * my * 2.0
    den0 = mx * mx + my * my
    lum = (num0 + c1) / (den0 + c1)
    num1 = filt(x * y) * 2.0
    den1 = filt(x * x + y * y)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(axis=(1, 2)).mean(axis=-1)


def gram_matrix(x):
    """SHM.py:1176-1180: einsum('bijc,bijd->bcd') / (H*W)."""
    return np.einsum('bijc,bijd->bcd', x, x) / float(x.shape[1] * x.shape[2])


def softmax_xent(labels, logits):
    """tf.nn.softmax_cross_entropy_with_logits -> [B] (SHM.py:695-713)."""
    z = logits - logits.max(axis=-1, keepdims=True)
    logsm = z - np.log(np.exp(z).sum(axis=-1, keepdims=True))
    return -(labels * logsm).sum(axis=-1)


def softmax_xent_backprop(labels, logits):
    """Second output of TF's SoftmaxCrossEntropyWithLogits kernel (tensorflow/core/kernels/xent_op.h):
    backprop = softmax(logits) - labels.  The registered gradient multiplies it by the incoming loss gradient
    (nn_grad.py, _SoftmaxCrossEntropyWithLogitsGrad), so this -- not sum(labels)*softmax - labels -- is d loss / d logits
    as executed, whatever the labels sum to."""
    z = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=-1, keepdims=True) - labels


def exp_decay_lr(lr0, step, decay_steps=10000, decay_rate=0.95):
    """ExponentialDecay(staircase=False) SHM.py:169-171."""
    return lr0 * decay_rate ** (step / decay_steps)


def adam_update(w, m, v, g, iterations, lr0, beta1, beta2, eps=1e-7):
    """clip_by_value(g,-1,1) (SHM.py:860,869) + Keras adam_v2.Adam._resource_apply_dense.
    `iterations` is the optimizer's counter BEFORE this apply (0 on the first step)."""
    g = np.clip(g, -1.0, 1.0)
    t = iterations + 1
    lr = exp_decay_lr(lr0, iterations)
    alpha = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = m + (g - m) * (1.0 - beta1)
    v = v + (g * g - v) * (1.0 - beta2)
    w = w - alpha * m / (np.sqrt(v) + eps)
    return w, m, v
