"""A/B of the DMA tap-GEMM variants on stride-2 forward convolutions (discriminator forward, Conv2DTranspose input gradient).
Usage: python tools/bench_s2.py [--dt bf16,f32]"""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops
from shmgan_amd._lib import ShmError

dts = ["bf16", "f32"]
if "--dt" in sys.argv:
    dts = sys.argv[sys.argv.index("--dt") + 1].split(",")
SHAPES = [(40, 256, 64, 128), (40, 128, 128, 256), (40, 64, 256, 512), (96, 128, 64, 128), (96, 64, 128, 256), (96, 32, 256, 512), (96, 16, 512, 1024)]
VARIANTS = ["auto", "dma128x128", "dma128x128_bk32", "dma128x128_nst4", "dma64x128", "dma256x128", "dma128x64"]
for dtn in dts:
    dt = torch.bfloat16 if dtn == "bf16" else torch.float32
    for n, h, cin, cout in SHAPES:
        x = torch.randn((n, h, h, cin), device="cuda").to(dt)
        w = torch.randn((3, 3, cin, cout), device="cuda") * 0.05
        wk = torch.zeros(9 * cout * cin, device="cuda", dtype=dt)
        ops.transpose_taps(w, wk, 9, cin, cout, cin)
        y = torch.empty((n, h // 2, h // 2, cout), device="cuda", dtype=dt)
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
        fn = lambda: ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, None, y, cout, n, h, h, cin, cout, 3, 2, 0.2, stats, 1e-6, scratch=scr)
        flops = 2.0 * n * (h // 2) ** 2 * 9 * cin * cout
        ok, times = [], {}
        for v in VARIANTS:
            ops.set_tuning("tapgemm.variant", v)
            try:
                fn(); torch.cuda.synchronize(); ok.append(v); times[v] = []
            except ShmError:
                pass
        for _ in range(5):
            for v in ok:
                ops.set_tuning("tapgemm.variant", v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record(); torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
        ops.set_tuning("reset", 0)
        print(f"{dtn:5s} n{n} h{h} {cin}->{cout} s2  " + "  ".join(f"{v} {statistics.median(times[v]):6.1f}us({flops / statistics.median(times[v]) / 1e6:5.0f})" for v in ok), flush=True)
