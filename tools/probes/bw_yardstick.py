"""Yardstick for the HBM-bound passes: what a plain 2-read / 1-write and a 1-read / 1-write elementwise kernel (torch's, used here only as
a measuring stick) reach on the tensors of the largest layer, beside shm_in_apply (1R + 1W) and shm_in_bwd_apply (2R + 1W)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from shmgan_amd import ops


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


import itertools
SB = [int(v) for v in sys.argv[1:]] or [4096, 8192, 16384, 32768]
for dt, sb in itertools.product((torch.float32, torch.bfloat16), SB):
    ops.set_tuning("elem.stream_blocks", sb)
    print("elem.stream_blocks", sb)
    n, h, c = 40, 256, 64
    es = 2 if dt == torch.bfloat16 else 4
    a = torch.randn((n, h, h, c), device="cuda").to(dt)
    g = torch.randn((n, h, h, c), device="cuda").to(dt)
    o = torch.empty_like(a)
    nb = a.numel() * es
    stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    beta = torch.zeros(c, device="cuda")
    red = torch.zeros(ops.GSUM_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
    dstage = torch.zeros(n * c, dtype=torch.float64, device="cuda")
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    rows = [("torch copy_ (1R+1W)", lambda: o.copy_(a), 2), ("torch add (2R+1W)", lambda: torch.add(a, g, out=o), 3),
            ("shm_in_apply (1R+1W)", lambda: ops.in_apply(a, c, stats, beta, o, c, n, h * h, c), 2),
            ("shm_in_bwd_apply (2R+1W)", lambda: ops.in_bwd_apply(g, c, None, 0, a, c, stats, beta, red, None, dstage, o, c, db, n, h, h, c, 0.2), 3)]
    for name, fn, k in rows:
        us = timeit(fn)
        print(f"{str(dt)[6:]:9s} {name:28s} {us:7.1f} us  {k * nb / us / 1e3:6.0f} GB/s", flush=True)
    us = timeit(lambda: ops.in_bwd_apply(g, c, None, 0, a, c, stats, beta, red, None, None, o, c, None, n, h, h, c, 0.2))
    print(f"{str(dt)[6:]:9s} {'shm_in_bwd_apply no dbias':28s} {us:7.1f} us  {3 * nb / us / 1e3:6.0f} GB/s", flush=True)
    ops.set_tuning("elem.interleave", 1)
    us = timeit(rows[-1][1])
    print(f"{str(dt)[6:]:9s} {'shm_in_bwd_apply interleaved':28s} {us:7.1f} us  {3 * nb / us / 1e3:6.0f} GB/s", flush=True)
    ops.set_tuning("reset", 0)
