"""Shared helpers for the parity tests (oracle side in float64 on the CPU)."""
import numpy as np
import torch

from oracle import step_torch as st


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy().astype(np.float64)


def t64(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def nchw(a):
    return t64(a).permute(0, 3, 1, 2)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def conv_ref(x, w_hwio, stride):
    return nhwc(st.conv2d_same(nchw(x), t64(w_hwio), stride))


def pad_c(x, c):
    """zero-pad the channel axis to c."""
    out = np.zeros(x.shape[:-1] + (c,), x.dtype)
    out[..., :x.shape[-1]] = x
    return out
