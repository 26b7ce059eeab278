"""A/B of the four-phase launches (Conv2DTranspose forward, stride-2 input gradient): fused tapgemm_phase4_kernel against
the DMA tap GEMM's four grid slices.  Usage: python tools/bench_phase4.py [--dt bf16|f32]"""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops
from shmgan_amd._lib import ShmError

dts = ["f32", "bf16"]
if "--dt" in sys.argv:
    dts = sys.argv[sys.argv.index("--dt") + 1].split(",")
# (kind, n, h_in, cin, cout): convT input map h_in -> 2 h_in; dgrad: dx map h_in (dy map h_in / 2), dx channels cin
SHAPES = [("convT", 40, 128, 128, 64), ("convT", 40, 64, 256, 128), ("convT", 40, 32, 512, 256), ("convT", 40, 16, 512, 512),
          ("convT", 8, 128, 128, 64), ("convT", 8, 64, 256, 128), ("convT", 8, 32, 512, 256), ("convT", 8, 16, 512, 512),
          ("dgrad", 96, 128, 64, 128), ("dgrad", 96, 64, 128, 256), ("dgrad", 96, 32, 256, 512), ("dgrad", 48, 128, 64, 128),
          ("dgrad", 48, 64, 128, 256), ("dgrad", 48, 32, 256, 512)]
VARIANTS = ["phase4", "dma128x128", "dma128x64", "dma64x128"]
for dtn in dts:
    dt = torch.bfloat16 if dtn == "bf16" else torch.float32
    for kind, n, h, cin, cout in SHAPES:
        if kind == "convT":
            x = torch.randn((n, h, h, cin), device="cuda").to(dt)
            w = (torch.randn((3, 3, cout, cin), device="cuda") * 0.05).to(dt)
            b = torch.randn(cout, device="cuda")
            y = torch.empty((n, 2 * h, 2 * h, cout), device="cuda", dtype=dt)
            fn = lambda: ops.conv2d_transpose_fwd(x, cin, w, b, y, cout, n, h, h, cin, cout, 0.2)
            flops = 2.0 * n * h * h * 9 * cin * cout
        else:
            dy = torch.randn((n, h // 2, h // 2, cout), device="cuda").to(dt)
            w = (torch.randn((3, 3, cin, cout), device="cuda") * 0.05).to(dt)
            dx = torch.empty((n, h, h, cin), device="cuda", dtype=dt)
            fn = lambda: ops.conv2d_dgrad(dy, cout, w, dx, None, cin, cin, 0, n, h, h, cin, cout, 3, 2)
            flops = 2.0 * n * (h // 2) * (h // 2) * 9 * cin * cout
        ok, times = [], {}
        for v in VARIANTS:
            ops.set_tuning("tapgemm.variant", v)
            try:
                fn()
                torch.cuda.synchronize()
                ok.append(v)
                times[v] = []
            except ShmError:
                pass
        for _ in range(5):
            for v in ok:
                ops.set_tuning("tapgemm.variant", v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
        ops.set_tuning("reset", 0)
        print(f"{dtn:5s} {kind:6s} n{n} h{h} {cin}->{cout}  " + "  ".join(f"{v} {statistics.median(times[v]):7.1f} us ({flops / statistics.median(times[v]) / 1e6:6.1f} TF)" for v in ok), flush=True)
