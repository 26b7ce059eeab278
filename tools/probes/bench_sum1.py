import sys; sys.path.insert(0, '/root/repo')
import torch
from shmgan_amd import ops
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dt in (torch.float32, torch.bfloat16):
    B, S, F = 8, 256, 64
    dz = torch.randn((5 * B, S, S, F), device="cuda").to(dt)
    weff = torch.randn((5, 9, F), device="cuda")
    out = torch.zeros((B, S, S, 1), device="cuda")
    print(dt, "G cyc stride1", timeit(lambda: ops.conv3x3_dgrad_sum1(dz, F, weff, out, 5, B, S, S, F, 1, 1)), "us")
    dzd = torch.randn((5 * B, S // 2, S // 2, F), device="cuda").to(dt)
    out5 = torch.zeros((5 * B, S, S, 1), device="cuda")
    print(dt, "D stride2 5B", timeit(lambda: ops.conv3x3_dgrad_sum1(dzd, F, weff[0], out5, 1, 5 * B, S, S, F, 2, 1)), "us")
