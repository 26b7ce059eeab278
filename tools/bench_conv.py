"""Micro-benchmark of the tap-GEMM family per shape and dtype: conv forward without / with the fused
InstanceNorm statistics, input gradient, weight gradient.  Prints us, TFLOP/s and algorithmic GB/s."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops

SHAPES = [(40, 256, 64, 64), (40, 256, 128, 64), (40, 128, 128, 128), (40, 64, 256, 256), (40, 32, 512, 512), (8, 256, 64, 64)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]


def timeit(fn, reps=5):
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
    es = 2 if dt == torch.bfloat16 else 4
    for n, h, cin, cout in SHAPES:
        x = torch.randn((n, h, h, cin), device="cuda").to(dt)
        dy = torch.randn((n, h, h, cout), device="cuda").to(dt)
        w = (torch.randn((3, 3, cin, cout), device="cuda") * 0.05)
        wk = torch.zeros(9 * cout * cin, device="cuda", dtype=dt)
        ops.transpose_taps(w, wk, 9, cin, cout, cin)
        wop = w.to(dt)
        y = torch.empty((n, h, h, cout), device="cuda", dtype=dt)
        dx = torch.empty((n, h, h, cin), device="cuda", dtype=dt)
        dw = torch.empty((3, 3, cin, cout), device="cuda")
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
        ws = torch.empty(ops.conv2d_wgrad_workspace(n, h, h, cin, cout, 3) // 4 + 1024, device="cuda")
        flops = 2.0 * n * h * h * 9 * cin * cout
        byts = es * n * h * h * (cin + cout)
        rows = [
            ("fwd", lambda: ops.conv2d_fwd(x, None, 0, cin, 0, wk, None, y, cout, n, h, h, cin, cout, 3, 1, 0.2)),
            ("fwd+stats", lambda: ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, None, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6)),
            ("fwd+stats16", lambda: ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, None, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)),
            ("dgrad", lambda: ops.conv2d_dgrad(dy, cout, wop, dx, None, cin, cin, 0, n, h, h, cin, cout, 3, 1)),
            ("wgrad", lambda: ops.conv2d_wgrad(x, None, 0, cin, 0, dy, cout, dw, n, h, h, cin, cin, cout, 3, 1, 0, ws)),
        ]
        for name, fn in rows:
            us = timeit(fn)
            print(f"{str(dt)[6:]:9s} n{n} h{h} {cin}->{cout} {name:10s} {us:8.1f} us {flops / us / 1e6:7.1f} TF  {byts / us / 1e3:6.0f} GB/s", flush=True)
