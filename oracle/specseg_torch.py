"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch-CPU, float64 by default) of the SpecSeg mask
network as the reference's train_step executes it, and of the specular loss it feeds.

Follows /root/reference/SpecSeg.py:27-98 (layer list; `predict` => Dropout inactive, BatchNormalization
in inference mode with Keras' default epsilon 1e-3), the call site ShmGANwithSSpecSeg.py:492
(`SpecSeg.predict(I90_Ych)`) and the Spec_loss terms ShmGANwithSSpecSeg.py:792-806.
Known answer pinned in tests/test_oracle.py: 1,942,801 parameters, 992 non-trainable
(/root/reference/SpecSeg_summary.txt:118-120).  PARITY UNPINNED beyond that: the reference's
checkpoint (specsegv3_chkpt.h5) and TensorFlow are not available, so values are checked against this
restatement only.  Nothing under shmgan_amd/ may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .step_torch import conv2d_same

WIDTHS = (16, 32, 64, 128, 256)
BN_EPS = 1e-3


def specseg_spec():
    """[(kind, shape)] in Keras get_weights() order: conv kernel HWIO + bias; bn gamma, beta,
    moving_mean, moving_variance; convT kernel [2,2,Cout,Cin] + bias."""
    out = []
    cin = 1
    for w in WIDTHS:                                   # SpecSeg.py:34-60
        out += [("conv", (3, 3, cin, w)), ("bias", (w,)), ("conv", (3, 3, w, w)), ("bias", (w,))]
        out += [("bn_gamma", (w,)), ("bn_beta", (w,)), ("bn_mean", (w,)), ("bn_var", (w,))]
        cin = w
    for w in WIDTHS[3::-1]:                            # SpecSeg.py:63-86
        out += [("convT", (2, 2, w, 2 * w)), ("bias", (w,))]
        out += [("conv", (3, 3, 2 * w, w)), ("bias", (w,)), ("conv", (3, 3, w, w)), ("bias", (w,))]
    out += [("conv", (1, 1, WIDTHS[0], 1)), ("bias", (1,))]      # SpecSeg.py:88
    return out


def init_specseg(seed=44, trained_like=True):
    """Random stand-in for the missing checkpoint.  trained_like: non-trivial biases and BN statistics so
    every term of the forward is exercised."""
    rng = np.random.default_rng(seed)
    ws = []
    for kind, s in specseg_spec():
        if kind in ("conv", "convT"):
            fan = s[0] * s[1] * (s[2] if kind == "conv" else s[3])
            ws.append(rng.normal(0.0, np.sqrt(2.0 / fan), s))
        elif kind == "bn_var":
            ws.append(rng.uniform(0.5, 1.5, s) if trained_like else np.ones(s))
        elif kind == "bn_gamma":
            ws.append(rng.uniform(0.8, 1.2, s) if trained_like else np.ones(s))
        else:
            ws.append(rng.normal(0.0, 0.1, s) if trained_like else np.zeros(s))
    return [w.astype(np.float32) for w in ws]


def specseg_forward(weights, x, dtype=torch.float64):
    """x [N,S,S,1] -> mask [N,S,S,1].  NHWC in/out (NCHW inside)."""
    W = [torch.as_tensor(np.asarray(w)).to(dtype) for w in weights]
    x = torch.as_tensor(np.asarray(x)).to(dtype).permute(0, 3, 1, 2)
    cv = lambda t: t.view(1, -1, 1, 1)
    it = iter(range(len(W)))

    def conv_relu(t):
        k, b = W[next(it)], W[next(it)]
        return torch.relu(conv2d_same(t, k, 1) + cv(b))

    def bn(t):
        g, be, mu, var = W[next(it)], W[next(it)], W[next(it)], W[next(it)]
        return (t - cv(mu)) * cv(g / torch.sqrt(var + BN_EPS)) + cv(be)

    def conv_t2(t):
        k, b = W[next(it)], W[next(it)]                 # [2,2,Cout,Cin]
        n, _, h, w_ = t.shape
        co = k.shape[2]                                 # out[2a+p, 2b+q, o] = sum_c x[a,b,c] k[p,q,o,c]
        y = torch.einsum('nchw,pqoc->nohpwq', t, k).reshape(n, co, 2 * h, 2 * w_)
        return y + cv(b)

    skips = []
    cur = x
    for l in range(5):
        cur = bn(conv_relu(conv_relu(cur)))
        if l < 4:
            skips.append(cur)
            cur = F.max_pool2d(cur, 2)
    for l in (3, 2, 1, 0):
        u = conv_t2(cur)
        cur = conv_relu(conv_relu(torch.cat([u, skips[l]], dim=1)))
    k, b = W[next(it)], W[next(it)]
    return torch.sigmoid(conv2d_same(cur, k, 1) + cv(b)).permute(0, 2, 3, 1).contiguous()


def spec_loss(cyc_yuv, ds_yuv, mask):
    """SHM.py:792-806.  cyc_yuv / ds_yuv: lists of 5 [B,S,S,3]; mask [B,S,S,1].
    Returns (Spec_loss, [5 terms])."""
    t = [torch.mean((c * mask - d * mask) ** 2) for c, d in zip(cyc_yuv, ds_yuv)]
    return (t[0] + t[1] + t[2] + t[3]) / 5.0 + t[4] * 5.0, t
