"""SpecSeg mask network (inference only) on the HIP kernels.

Mirrors /root/reference/SpecSeg.py:27-98 as `train_step` uses it (`self.SpecSeg.predict(I90_Ych)`,
SHM.py:492; test.py:221): a U-Net of Conv2D(3x3, relu) pairs with inference-mode
BatchNormalization after each encoder pair, MaxPooling2D(2), Conv2DTranspose(2x2, stride 2) +
Concatenate([up, skip]) in the decoder and a Conv2D(1, 1x1, sigmoid) head; Dropout layers are
inactive under `predict`.  Widths are fixed (16..256) whatever the GAN's filter_size.

The reference loads `specsegv3_chkpt.h5`, which is not part of the mount: `init_random()` gives the
Keras initialisers of SpecSeg.py instead, `set_weights()` takes `model.get_weights()` of a trained
Keras SpecSeg (variable order and layouts preserved: HWIO kernels, Conv2DTranspose
[kh,kw,Cout,Cin], BatchNormalization gamma/beta/moving_mean/moving_variance).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops

WIDTHS = (16, 32, 64, 128, 256)
BN_EPS = 1e-3                      # Keras BatchNormalization default (SpecSeg.py:37 passes none)
PAD_C = 16


def specseg_variables():
    """[(keras_name, kind, shape)] in `model.get_weights()` order (SpecSeg_summary.txt)."""
    out = []
    ci = [0]

    def conv(cin, cout, k=3):
        n = "conv2d" if ci[0] == 0 else f"conv2d_{ci[0]}"
        ci[0] += 1
        out.append((n + "/kernel", "conv", (k, k, cin, cout)))
        out.append((n + "/bias", "bias", (cout,)))

    cin = 1
    for l, w in enumerate(WIDTHS):
        conv(cin, w)
        conv(w, w)
        bn = "batch_normalization" if l == 0 else f"batch_normalization_{l}"
        for p in ("gamma", "beta", "moving_mean", "moving_variance"):
            out.append((f"{bn}/{p}", "bn", (w,)))
        cin = w
    for j, l in enumerate((3, 2, 1, 0)):
        w = WIDTHS[l]
        t = "conv2d_transpose" if j == 0 else f"conv2d_transpose_{j}"
        out.append((t + "/kernel", "convT", (2, 2, w, 2 * w)))
        out.append((t + "/bias", "bias", (w,)))
        conv(2 * w, w)
        conv(w, w)
    conv(WIDTHS[0], 1, k=1)
    return out


class SpecSeg:
    name = "SpecSeg"
    trainable = False

    def __init__(self, image_size, device, arena):
        assert image_size % 16 == 0, "SpecSeg needs image_size % 16 == 0 (four 2x2 pools)"
        self.S, self.dev, self.arena = image_size, device, arena
        self.spec = specseg_variables()
        sizes = [int(np.prod(s)) for _, _, s in self.spec]
        self.n = sum(sizes)
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=device)
        self.vars, off = [], 0
        for (_, _, s), z in zip(self.spec, sizes):
            self.vars.append(self.flat[off:off + z].view(s))
            off += z
        self.index = {n: i for i, (n, _, _) in enumerate(self.spec)}
        self.wk = {}
        for i, (n, kind, s) in enumerate(self.spec):
            if kind == "conv" and s[3] > 1:
                k, _, cin, cout = s
                self.wk[i] = torch.zeros(k * k * cout * _pad16(cin), dtype=torch.float32, device=device)
        self.weights_dirty = True

    # ---- parameters ---------------------------------------------------------------------
    def count_params(self):
        return self.n

    def get_weights(self):
        return [v.detach().cpu().numpy().copy() for v in self.vars]

    def set_weights(self, arrays):
        assert len(arrays) == len(self.vars)
        for v, a in zip(self.vars, arrays):
            a = torch.as_tensor(np.asarray(a, dtype=np.float32))
            assert tuple(a.shape) == tuple(v.shape), (a.shape, v.shape)
            v.copy_(a)
        self.weights_dirty = True

    def init_random(self, seed=44):
        """Keras initialisers of SpecSeg.py: RandomNormal(0, 0.05) on the 3x3 kernels, glorot_uniform on
        the Conv2DTranspose / head kernels, zeros for biases, BN gamma 1, beta 0, mean 0, variance 1."""
        rng = np.random.default_rng(seed)
        ws = []
        for n, kind, s in self.spec:
            if kind == "conv" and s[0] == 3:
                ws.append(rng.normal(0.0, 0.05, s).astype(np.float32))
            elif kind in ("conv", "convT"):
                rf = s[0] * s[1]
                lim = np.sqrt(6.0 / (rf * s[2] + rf * s[3]))
                ws.append(rng.uniform(-lim, lim, s).astype(np.float32))
            elif kind == "bn":
                ws.append(np.ones(s, np.float32) if n.endswith(("gamma", "moving_variance")) else np.zeros(s, np.float32))
            else:
                ws.append(np.zeros(s, np.float32))
        self.set_weights(ws)
        return self

    def prepare_weights(self):
        if not self.weights_dirty:
            return
        for i, wk in self.wk.items():
            k, _, cin, cout = self.spec[i][2]
            ops.transpose_taps(self.vars[i], wk, k * k, cin, cout, _pad16(cin))
        self.weights_dirty = False

    # ---- forward ------------------------------------------------------------------------
    def _conv(self, tag, i, x, x2, c1, ldx, ldx2, n, h):
        k, _, cin, cout = self.spec[i][2]
        y = self.arena.get(f"{tag}/c{i}", (n, h, h, cout))
        ops.conv2d_fwd(x, x2, c1, ldx, ldx2, self.wk[i], self.vars[i + 1], y, cout, n, h, h, _pad16(cin), cout, k, 1, 0.0,
                       cin_real=cin)
        return y

    def forward_plane(self, src, ldsrc, c0, n, tag="specseg"):
        """Mask of channel c0 of `src` ([n,S,S,ldsrc]); returns [n,S,S,1] in (0,1)."""
        S, A = self.S, self.arena
        self.prepare_weights()
        x16 = A.get(f"{tag}/x16", (n, S, S, PAD_C))
        ops.pack_channels(src, ldsrc, c0, 1, x16, PAD_C, n * S * S)
        cur, ld, h = x16, PAD_C, S
        i = 0
        skips = []
        for l, w in enumerate(WIDTHS):
            a = self._conv(tag, i, cur, None, 0, ld, 0, n, h)
            b = self._conv(tag, i + 2, a, None, 0, w, 0, n, h)
            c = A.get(f"{tag}/bn{l}", (n, h, h, w))
            g, be, mu, var = self.vars[i + 4:i + 8]
            ops.bn_apply(b, w, g, be, mu, var, BN_EPS, c, w, n * h * h, w)
            i += 8
            if l < 4:
                skips.append((c, w, h))
                p = A.get(f"{tag}/p{l}", (n, h // 2, h // 2, w))
                ops.maxpool2_fwd(c, w, p, w, n, h, h, w)
                cur, ld, h = p, w, h // 2
            else:
                cur, ld = c, w
        for l in (3, 2, 1, 0):
            w = WIDTHS[l]
            u = A.get(f"{tag}/u{l}", (n, 2 * h, 2 * h, w))
            ops.conv2d_transpose2x2_fwd(cur, ld, self.vars[i], self.vars[i + 1], u, w, n, h, h, 2 * w, w, 1.0)
            i += 2
            h *= 2
            skip, sw, sh = skips[l]
            assert sh == h and sw == w
            a = self._conv(tag, i, u, skip, w, w, w, n, h)            # Concatenate([up, skip])
            b = self._conv(tag, i + 2, a, None, 0, w, 0, n, h)
            i += 4
            cur, ld = b, w
        y = A.get(f"{tag}/mask", (n, S, S, 1))
        ops.head_sigmoid_fwd(cur, ld, self.vars[i], self.vars[i + 1], y, n * S * S, ld)
        return y

    def predict(self, x, verbose=0):
        """Keras-style `SpecSeg.predict(I90_Ych, verbose=0)` on a [N,S,S,1] tensor (SHM.py:492)."""
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        x = x.to(self.dev, torch.float32).contiguous()
        assert x.dim() == 4 and x.shape[1] == self.S and x.shape[2] == self.S and x.shape[3] == 1
        return self.forward_plane(x, 1, 0, x.shape[0], tag="specseg/predict")

    __call__ = predict

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        for (n, _, s), v in zip(self.spec, self.vars):
            print_fn(f"{n:40s} {tuple(s)}  {v.numel()}")
        nt = sum(v.numel() for (n, k, _), v in zip(self.spec, self.vars) if n.endswith(("moving_mean", "moving_variance")))
        print_fn(f"Total params: {self.n:,}")
        print_fn(f"Trainable params: {self.n - nt:,}")
        print_fn(f"Non-trainable params: {nt:,}")


def _pad16(c):
    return (c + 15) // 16 * 16
