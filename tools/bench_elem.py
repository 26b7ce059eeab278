"""Micro-benchmark of the HBM-bound elementwise kernels at the largest layer shape (n=40, 256x256, 64 ch):
achieved GB/s = algorithmic bytes / HIP-event time."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
h, c = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 64)
only = sys.argv[4] if len(sys.argv) > 4 else ""        # substring filter on the row names
dev = "cuda"


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


for dt in (torch.float32, torch.bfloat16):
    es = 2 if dt == torch.bfloat16 else 4
    nel = n * h * h * c
    a = torch.randn((n, h, h, c), device=dev).to(dt)
    g = torch.randn((n, h, h, c), device=dev).to(dt)
    g2 = torch.randn((n, h // 2, h // 2, c), device=dev).to(dt)
    out = torch.empty_like(a)
    stats = torch.empty(n * c * 2, dtype=torch.float64, device=dev)
    red = torch.zeros(n * c * 3, dtype=torch.float64, device=dev)
    lred = torch.zeros(64 * c, dtype=torch.float64, device=dev)
    beta = torch.zeros(c, device=dev)
    db = torch.zeros(c, dtype=torch.float64, device=dev)
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    rows = [
        ("in_stats", lambda: ops.in_stats(a, c, stats, n, h * h, c, 1e-6), es),
        ("in_apply", lambda: ops.in_apply(a, c, stats, beta, out, c, n, h * h, c), 2 * es),
        ("in_bwd (no g2, dbias)", lambda: ops.in_bwd(g, c, None, 0, a, c, stats, red, out, c, db, n, h, h, c, 0.2), 5 * es),
        ("in_bwd (no g2, no dbias)", lambda: ops.in_bwd(g, c, None, 0, a, c, stats, red, out, c, None, n, h, h, c, 0.2), 5 * es),
        ("in_bwd (g2, dbias)", lambda: ops.in_bwd(g, c, g2, c, a, c, stats, red, out, c, db, n, h, h, c, 0.2), 5.5 * es),
        ("lrelu_bwd (dbias)", lambda: ops.lrelu_bwd(g, c, a, c, out, c, db, n * h * h, c, 0.2, lred), 3 * es),
        ("lrelu_bwd (no dbias)", lambda: ops.lrelu_bwd(g, c, a, c, out, c, None, n * h * h, c, 0.2), 3 * es),
        ("avgpool2", lambda: ops.avgpool2_fwd(a, c, g2, c, n, h, h, c), 1.25 * es),
    ]
    gs = ops.GSUM_SLOTS
    gred = torch.zeros(gs * n * c * 2, dtype=torch.float64, device=dev)
    gredp = torch.zeros(gs * n * c * 2, dtype=torch.float64, device=dev)
    dstage = torch.zeros(n * c, dtype=torch.float64, device=dev)
    rows += [
        ("in_bwd_apply (gsum, dbias)", lambda: ops.in_bwd_apply(g, c, None, 0, a, c, stats, beta, gred, None, dstage, out, c, db, n, h, h, c, 0.2), 3 * es),
        ("in_bwd_apply (gsum, g2)", lambda: ops.in_bwd_apply(g, c, g2, c, a, c, stats, beta, gred, gredp, dstage, out, c, db, n, h, h, c, 0.2), 3.25 * es),
    ]
    for name, fn, bpe in rows:
        if only not in name:
            continue
        us = timeit(fn)
        print(f"{str(dt):16s} {name:28s} {us:8.1f} us  {nel * bpe / us / 1e3:7.0f} GB/s")
