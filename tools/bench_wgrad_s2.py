"""A/B of the stride-2 3x3 weight gradient: halo form (wgrad_halo_kernel<0, true>) against the generic kernel (wgrad.variant 3), fp32, on the
step's layer shapes (generator encoder / Conv2DTranspose at n = 40, discriminator at n = 96).  usage: bench_wgrad_s2.py [n,h,cin,cout ...]"""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops

SHAPES = [(40, 256, 64, 128), (40, 128, 128, 256), (40, 64, 256, 512), (40, 32, 512, 512), (96, 128, 64, 128), (96, 64, 128, 256), (96, 32, 256, 512),
          (96, 16, 512, 1024), (8, 256, 64, 128), (8, 64, 256, 512)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for n, h, cin, cout in SHAPES:
    x = torch.randn((n, h, h, cin), device="cuda")
    dy = torch.randn((n, h // 2, h // 2, cout), device="cuda")
    dw = torch.empty((3, 3, cin, cout), device="cuda")
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, h // 2, h // 2, cin, cout, 3) // 4 + 1024, device="cuda")
    fn = lambda: ops.conv2d_wgrad(x, None, 0, cin, 0, dy, cout, dw, n, h, h, cin, cin, cout, 3, 2, 0, ws)
    flops = 2.0 * n * (h // 2) ** 2 * 9 * cin * cout
    times, names = {0: [], 3: []}, {}
    for _ in range(5):
        for wv in (0, 3):
            ops.set_tuning("wgrad.variant", wv)
            fn()
            names[wv] = ops.last_kernel()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[wv].append(e0.elapsed_time(e1) / 4 * 1e3)
    ops.set_tuning("reset", 0)
    t0, t3 = statistics.median(times[0]), statistics.median(times[3])
    print(f"n{n} h{h} {cin}x{cout} s2: {names[0]} {t0:7.1f} us ({flops / t0 / 1e6:5.1f} TF) | {names[3]} {t3:7.1f} us ({flops / t3 / 1e6:5.1f} TF) | {100 * (t0 / t3 - 1):+5.1f} %",
          flush=True)
