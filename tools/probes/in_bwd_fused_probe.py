"""Times shm_in_bwd (bf16) with the one-pass form on and off on the step's shapes.  python tools/probes/in_bwd_fused_probe.py [--pool]"""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from shmgan_amd import ops

BF = torch.bfloat16
pool = "--pool" in sys.argv
CASES = [0, 1]
if "--big" in sys.argv or "--max512" in sys.argv:
    ops.set_tuning("elem.fused_max_slices", 512)
shapes = [(20, 256, 128), (20, 128, 256), (20, 64, 512), (4, 256, 128)] if "--big" in sys.argv else [(40, 256, 64), (40, 128, 128), (40, 64, 256), (40, 32, 512), (8, 256, 64), (16, 128, 64), (16, 64, 128), (16, 32, 256), (16, 16, 512)]
for n, h, c in shapes:
    a = torch.randn(n, h, h, c, device="cuda").to(BF)
    g = torch.randn(n, h, h, c, device="cuda").to(BF)
    g2 = torch.randn(n, h // 2, h // 2, c, device="cuda").to(BF) if pool else None
    stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    dz = torch.empty_like(a)
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    out = []
    for fb in CASES:
        ops.set_tuning("elem.fused_bwd", fb)
        for _ in range(3):
            ops.in_bwd(g, c, g2, c if pool else 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=scratch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.in_bwd(g, c, g2, c if pool else 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=scratch)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    mb = a.numel() * 2 / 1e6
    print(f"n={n:3d} h={h:3d} c={c:3d} tensor {mb:7.1f} MB  two-pass {out[0]:7.1f} us ({5 * mb / out[0]:5.2f} TB/s of 5 passes)  "
          f"one pass {out[1]:7.1f} us ({3 * mb / out[1]:5.2f} TB/s of 3 passes)  timeout word {float(scratch[-1])}", flush=True)
