"""Bitwise run-to-run reproducibility of one convolution entry point under a forced tap-GEMM variant.
python tools/probes/conv_repeat_probe.py [--dt bf16|f32] [--variants a,b] n,h,cin,cout ..."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from shmgan_amd import ops

args = sys.argv[1:]
dts, variants, shapes = ["bf16"], ["halo128_st"], []
i = 0
while i < len(args):
    if args[i] == "--dt":
        dts = args[i + 1].split(","); i += 2
    elif args[i] == "--variants":
        variants = args[i + 1].split(","); i += 2
    else:
        shapes.append(tuple(int(v) for v in args[i].split(","))); i += 1
for dtn in dts:
    dt = torch.bfloat16 if dtn == "bf16" else torch.float32
    for n, h, cin, cout in shapes:
        torch.manual_seed(0)
        x = torch.randn((n, h, h, cin), device="cuda").to(dt)
        w = torch.randn((3, 3, cin, cout), device="cuda") * 0.05
        wk = torch.zeros(9 * cout * cin, device="cuda", dtype=dt)
        ops.transpose_taps(w, wk, 9, cin, cout, cin)
        b = torch.randn(cout, device="cuda")
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
        for v in variants:
            ops.set_tuning("tapgemm.variant", v)
            ref, bad = None, 0
            for r in range(30):
                y = torch.empty((n, h, h, cout), device="cuda", dtype=dt)
                ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, b, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)
                torch.cuda.synchronize()
                if ref is None:
                    ref = y.clone()
                elif not torch.equal(ref, y):
                    bad += 1
                    d = (ref.float() - y.float()).abs()
                    if bad <= 3:
                        idx = torch.nonzero(d > 0)
                        print(f"   rep {r}: {int((d > 0).sum())} elements differ, max {float(d.max()):.4g}; first at {idx[0].tolist()} last {idx[-1].tolist()}", flush=True)
            print(f"{dtn} n{n} h{h} {cin}->{cout} {v}: {bad} of 29 repetitions differ from the first ({ops.last_kernel()})", flush=True)
        ops.set_tuning("reset", 0)
