"""A/B of the tap-GEMM variants on one layer shape: every eligible variant is forced through shm_set_tuning and timed in
interleaved rounds inside ONE process (median / min over rounds), forward conv + fused InstanceNorm statistics and the
input gradient.  Usage: python tools/bench_variants.py [--dt bf16|f32] [--variants a,b,...] n,h,cin,cout [...]"""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops
from shmgan_amd._lib import ShmError

args = sys.argv[1:]
dts = ["bf16", "f32"]
variants = None
shapes = []
i = 0
while i < len(args):
    if args[i] == "--dt":
        dts = args[i + 1].split(",")
        i += 2
    elif args[i] == "--variants":
        variants = args[i + 1].split(",")
        i += 2
    else:
        shapes.append(tuple(int(v) for v in args[i].split(",")))
        i += 1
shapes = shapes or [(40, 256, 64, 64), (8, 256, 64, 64), (40, 128, 128, 128)]
variants = variants or ["auto", "wreg", "halo64", "halo128", "dma128x64", "dma128x128", "dma64x128", "dma256x64"]
ROUNDS, REPS = 5, 4


def force(v):
    """a variant name, or "knob:value" = automatic dispatch with tapgemm.<knob> set (e.g. wreg16:1 / wreg16:2: the two bf16 weights-in-registers kernels)"""
    ops.set_tuning("reset", 0)
    if ":" in v:
        k, val = v.split(":")
        ops.set_tuning("tapgemm." + k, int(val))
    else:
        ops.set_tuning("tapgemm.variant", v)


for dtn in dts:
    dt = torch.bfloat16 if dtn == "bf16" else torch.float32
    es = 2 if dtn == "bf16" else 4
    for n, h, cin, cout in shapes:
        x = torch.randn((n, h, h, cin), device="cuda").to(dt)
        dy = torch.randn((n, h, h, cout), device="cuda").to(dt)
        w = torch.randn((3, 3, cin, cout), device="cuda") * 0.05
        wk = torch.zeros(9 * cout * cin, device="cuda", dtype=dt)
        ops.transpose_taps(w, wk, 9, cin, cout, cin)
        wop = w.to(dt)
        b = torch.randn(cout, device="cuda")
        y = torch.empty((n, h, h, cout), device="cuda", dtype=dt)
        dx = torch.empty((n, h, h, cin), device="cuda", dtype=dt)
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
        flops = 2.0 * n * h * h * 9 * cin * cout
        byts = es * n * h * h * (cin + cout)
        calls = {"fwd+stats": lambda: ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, b, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr),
                 "dgrad": lambda: ops.conv2d_dgrad(dy, cout, wop, dx, None, cin, cin, 0, n, h, h, cin, cout, 3, 1)}
        for cname, fn in calls.items():
            ok, times, syms = [], {}, {}
            for v in variants:
                force(v)
                try:
                    fn()
                    torch.cuda.synchronize()
                    ok.append(v)
                    syms[v] = ops.last_kernel()
                    times[v] = []
                except ShmError:
                    pass
            for _ in range(ROUNDS):
                for v in ok:
                    force(v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(REPS):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[v].append(e0.elapsed_time(e1) / REPS * 1e3)
            for v in ok:
                med, mn = statistics.median(times[v]), min(times[v])
                print(f"{dtn:5s} n{n} h{h} {cin}->{cout} {cname:10s} {v:12s} {med:8.1f} us (min {mn:7.1f})  {flops / med / 1e6:7.1f} TF  "
                      f"{byts / med / 1e3:6.0f} GB/s  {syms[v]}", flush=True)
            ops.set_tuning("reset", 0)
