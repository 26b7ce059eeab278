"""The north-star block (conv3x3 64 -> 64 + bias + LeakyReLU + IN statistics, bf16) a few times, and its input gradient: the target of one-kernel
rocprofv3 passes (tools/pmc_dump.py).  python tools/probes/ns_once.py [n,h] [reps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from shmgan_amd import ops

n, h = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40,256").split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = 64
dt = torch.bfloat16
x = torch.randn((n, h, h, c), device="cuda").to(dt)
w = torch.randn((3, 3, c, c), device="cuda") * 0.05
wk = torch.zeros(9 * c * c, device="cuda", dtype=dt)
ops.transpose_taps(w, wk, 9, c, c, c)
b = torch.randn(c, device="cuda")
y = torch.empty((n, h, h, c), device="cuda", dtype=dt)
stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
scr = torch.zeros(ops.STATS_SLOTS * n * c * 2, dtype=torch.float64, device="cuda")
for _ in range(reps):
    ops.conv2d_in_fwd(x, None, 0, c, 0, wk, b, y, c, n, h, h, c, c, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    ops.conv2d_dgrad(y, c, w.to(dt), x, None, c, c, 0, n, h, h, c, c, 3, 1)
torch.cuda.synchronize()
print(ops.last_kernel())
