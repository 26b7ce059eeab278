"""Time the discriminator's first layer (csrc/conv_rgb.hip) at the step's size: forward (+ InstanceNorm sums) and weight gradient, compact layout
against the staging layout (generic kernels), both dtypes.   python tools/probes/bench_rgb.py [n] [S]"""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from shmgan_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cout = 64


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dt in (torch.float32, torch.bfloat16):
    esz = 4 if dt == torch.float32 else 2
    kpad = 64 // esz
    w = torch.randn((3, 3, 3, cout), device="cuda") * 0.3
    wk = torch.zeros(9 * cout * kpad, device="cuda", dtype=dt)
    ops.transpose_taps(w, wk, 9, 3, cout, kpad)
    y = torch.empty((n, S // 2, S // 2, cout), device="cuda", dtype=dt)
    dz = torch.randn((n, S // 2, S // 2, cout), device="cuda").to(dt)
    stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
    scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
    dw = torch.empty((3, 3, 3, cout), device="cuda")
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, S // 2, S // 2, 3, cout, 3) // 4 + 1024, device="cuda")
    for pitch in (16 // esz, kpad):
        x = torch.zeros((n, S, S, pitch), device="cuda", dtype=dt)
        x[..., :3] = torch.randn((n, S, S, 3), device="cuda").to(dt)
        tf = timed(lambda: ops.conv2d_in_fwd(x, None, 0, pitch, 0, wk, None, y, cout, n, S, S, kpad, cout, 3, 2, 0.2, stats, 1e-6, scratch=scr))
        kf = ops.last_kernel()
        tw = timed(lambda: ops.conv2d_wgrad(x, None, 0, pitch, 0, dz, cout, dw, n, S, S, 3, kpad, cout, 3, 2, 0, ws))
        kw = ops.last_kernel()
        out_b = y.numel() * esz
        x_b = n * S * S * 16
        print(f"{str(dt).split('.')[-1]:9s} pitch {pitch:2d}  fwd {tf:7.1f} us ({(out_b + x_b) / tf / 1e6:5.2f} TB/s of out + compact x) {kf.split('<')[0]}   "
              f"wgrad+reduce {tw:7.1f} us ({(out_b + x_b) / tw / 1e6:5.2f} TB/s) {kw.split('<')[0]}", flush=True)
