"""Phase stamps of tapgemm_wreg16_bf16_kernel (timing-only build with -DSHM_ABL_STAMP, loaded through SHM_LIB_PATH): cycles per patch and wave
between the phase boundaries of the patch loop, for one block in the middle of the grid.  The stamped build dumps over the bias vector."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from shmgan_amd import ops

n, h, cin, cout = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40,256,64,64").split(","))
dt = torch.bfloat16
x = torch.randn((n, h, h, cin), device="cuda").to(dt)
w = torch.randn((3, 3, cin, cout), device="cuda") * 0.05
wk = torch.zeros(9 * cout * cin, device="cuda", dtype=dt)
ops.transpose_taps(w, wk, 9, cin, cout, cin)
b = torch.zeros(cout, device="cuda")
y = torch.empty((n, h, h, cout), device="cuda", dtype=dt)
stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
ops.set_tuning("tapgemm.variant", "wreg")
for _ in range(5):
    b.zero_()
    ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, b, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    torch.cuda.synchronize()
print(ops.last_kernel())
d = b.view(torch.int32).cpu().numpy().reshape(8, 8)
names = ["top barrier", "DMA issue", "MFMA loop", "epilogue", "staging barrier", "store issue", "halo wait"]
npatch = d[0, 7]
print("patches per block:", npatch)
print("wave  " + "  ".join(f"{s:>15s}" for s in names) + "            total")
for wv in range(8):
    per = d[wv, :7] / max(npatch, 1)
    print(f"{wv:4d}  " + "  ".join(f"{v:15.0f}" for v in per) + f"  {per.sum():15.0f}")
