"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the device's step-level random draws (shm_randn / shm_keep_mask in
shmgan_amd/csrc/color.hip): Philox-4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11;
constants of Random123 philox.h), counter = (i_lo, i_hi, stream, kind) for the i-th group of four outputs, key = the 64-bit
seed; uniforms from the top 24 bits, normals by Box-Muller on two pairs.  Pinned by Random123's known-answer vectors
(tests/test_oracle.py).  The reference draws its noise / dropout masks inside Keras layers (SHM.py:352, 363) from TensorFlow's
generator: there is nothing of the reference to match bit for bit here, only the distributions."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c, k):
    """c: four uint64 arrays (32-bit values) of equal shape, k: two; returns four uint64 arrays."""
    c = [np.asarray(x, dtype=np.uint64) for x in c]
    k = [np.asarray(x, dtype=np.uint64) for x in k]
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [((p1 >> _S32) ^ c[1] ^ k[0]) & _MASK, p1 & _MASK, ((p0 >> _S32) ^ c[3] ^ k[1]) & _MASK, p0 & _MASK]
        k = [(k[0] + _W0) & _MASK, (k[1] + _W1) & _MASK]
    return c


def _words(n, seed, stream, kind):
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    z = np.zeros_like(g)
    return philox4x32_10([g & _MASK, g >> _S32, z + np.uint64(stream), z + np.uint64(kind)],
                         [np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)])


def keep_mask(n, rate, seed, stream):
    r = np.stack(_words(n, seed, stream, 1), axis=1).reshape(-1)[:n]
    u = (r >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(rate)).astype(np.float32)


def randn(n, stddev, seed, stream):
    r = [w.astype(np.uint64) for w in _words(n, seed, stream, 0)]
    out = np.empty((len(r[0]), 4), np.float64)
    for p in range(2):
        u1 = ((r[2 * p] >> np.uint64(8)).astype(np.float64) + 1.0) / 16777216.0
        u2 = (r[2 * p + 1] >> np.uint64(8)).astype(np.float64) / 16777216.0
        rad = np.sqrt(-2.0 * np.log(u1)) * stddev
        out[:, 2 * p] = rad * np.cos(2.0 * np.pi * u2)
        out[:, 2 * p + 1] = rad * np.sin(2.0 * np.pi * u2)
    return out.reshape(-1)[:n]
