"""Sweep of "elem.apply_blocks" on shm_in_bwd_apply (with / without the bias gradient) at the largest layer shape."""
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


for dt in (torch.float32, torch.bfloat16):
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
    red3 = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    dstage = torch.zeros(n * c, dtype=torch.float64, device="cuda")
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    for blocks in (2048, 4096, 8192, 16384, 32768):
        ops.set_tuning("elem.apply_blocks", blocks)
        t1 = timeit(lambda: ops.in_bwd_apply(g, c, None, 0, a, c, stats, beta, red, None, dstage, o, c, db, n, h, h, c, 0.2))
        t2 = timeit(lambda: ops.in_bwd_apply(g, c, None, 0, a, c, stats, beta, red, None, None, o, c, None, n, h, h, c, 0.2))
        t3 = timeit(lambda: ops.in_bwd(g, c, None, 0, a, c, stats, red3, o, c, db, n, h, h, c, 0.2))
        print(f"{str(dt)[6:]:9s} apply_blocks {blocks:6d}: in_bwd_apply {t1:6.1f} us {3 * nb / t1 / 1e3:5.0f} GB/s | no dbias {t2:6.1f} us {3 * nb / t2 / 1e3:5.0f} GB/s"
              f" | in_bwd (reduce + apply) {t3:6.1f} us {5 * nb / t3 / 1e3:5.0f} GB/s", flush=True)
