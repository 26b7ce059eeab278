"""A/B of the one-pass bf16 InstanceNorm-backward kernels (round 6): in_bwd_fused8_kernel (g and a held) against in_bwd_fusedg_kernel (g held, a streamed
twice) and its register-budget variants, on the step's shapes; also checks the outputs against each other.  python tools/probes/in_bwd_fusedg_ab.py [--big]"""
import sys
import torch

sys.path.insert(0, ".")
from shmgan_amd import ops

BF = torch.bfloat16
shapes = [(20, 512, 64), (20, 256, 128), (4, 512, 64)] if "--big" in sys.argv else [(40, 256, 64), (40, 128, 128), (40, 64, 256), (40, 32, 512), (8, 256, 64), (8, 128, 128), (160, 256, 64)]
cases = [("two-pass", dict(fused_bwd=0)), ("fused8", dict(fused_hold=1)), ("fusedg<2,2,4>", dict(fused_hold=2, fused_gvariant=0)), ("fusedg<8,8,3>", dict(fused_hold=2, fused_gvariant=1))]
for n, h, c in shapes:
    a = (torch.randn(n, h, h, c, device="cuda") * 1.5 + 0.4).to(BF)
    g = (torch.randn(n, h, h, c, device="cuda") + 3.0).to(BF)
    stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    scratch = torch.zeros(ops.in_bwd_fused_doubles(n, h * h, c), dtype=torch.float64, device="cuda")
    ref = None
    line = f"n={n:3d} h={h:3d} c={c:3d} ({a.numel() * 2 / 1e6:6.1f} MB): "
    for name, knobs in cases:
        ops.set_tuning("reset", 0)
        for k, v in knobs.items():
            ops.set_tuning("elem." + k, v)
        dz = torch.zeros_like(a)
        db = torch.zeros(c, dtype=torch.float64, device="cuda")
        ops.in_bwd(g, c, None, 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=scratch)
        kern = ops.last_kernel()
        torch.cuda.synchronize()
        if ref is None:
            ref = (dz.double(), db.clone())
        err = float((dz.double() - ref[0]).norm() / ref[0].norm())
        errb = float((db - ref[1]).norm() / ref[1].norm())
        dzt = torch.empty_like(a)
        for _ in range(3):
            ops.in_bwd(g, c, None, 0, a, c, stats, red, dzt, c, db, n, h, h, c, 0.2, fused=scratch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.in_bwd(g, c, None, 0, a, c, stats, red, dzt, c, db, n, h, h, c, 0.2, fused=scratch)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        tag = "" if name == "two-pass" or name.split("<")[0] in kern else f" [ran {kern}]"
        line += f"{name} {us:6.1f} us ({3 * a.numel() * 2 / us / 1e6:4.2f} TB/s; dz {err:.1e} db {errb:.1e}){tag} | "
    print(line, "timeout word", int(scratch[-1:].view(torch.int64).item()), flush=True)
