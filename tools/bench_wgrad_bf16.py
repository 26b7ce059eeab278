"""A/B of the bf16 3x3 weight gradient on the step's layer shapes (S=256: generator passes at n = 40 / 8, discriminator at n = 96;
--s512: BASELINE configs[3], n = 20 / 4 / 48): the eight-wave 64 x 128 block (wgrad_halo8_bf16_kernel, default where cout >= 128) against the
four-wave kernels ("wgrad.bf16_wide" = 1: wgrad_halo_bf16_kernel<4> at unit stride, wgrad_bf16_kernel<9> at stride 2), launch + slab
reduce, interleaved rounds in one process.  usage: bench_wgrad_bf16.py [--s512] [--blocks T] [--ab-blocks A,B] [n,h,cin,cout,stride ...]
--ab-blocks A,B: compare two split-K block targets under the default dispatch instead."""
import statistics
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops

args = [a for a in sys.argv[1:] if not a.startswith("--")]
s512 = "--s512" in sys.argv
blocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 0
if blocks:
    args = [a for a in args if a != str(blocks)]
ab = [int(v) for v in sys.argv[sys.argv.index("--ab-blocks") + 1].split(",")] if "--ab-blocks" in sys.argv else None
if ab:
    args = [a for a in args if a != sys.argv[sys.argv.index("--ab-blocks") + 1]]
S, (ng, n1, nd) = (512, (20, 4, 48)) if s512 else (256, (40, 8, 96))
# unit stride: generator blocks with >= 128 output channels; stride 2: Conv2DTranspose (roles swapped: x = its output gradient) and discriminator
SHAPES = [(ng, S // 2, 64, 128, 1), (ng, S // 2, 128, 128, 1), (ng, S // 4, 128, 256, 1), (ng, S // 4, 256, 256, 1), (ng, S // 8, 256, 512, 1), (ng, S // 8, 512, 512, 1),
          (ng, S // 8, 1024, 512, 1), (ng, S // 4, 512, 256, 1), (ng, S // 2, 256, 128, 1), (n1, S // 2, 128, 128, 1), (n1, S // 8, 512, 512, 1),
          (ng, S, 64, 128, 2), (ng, S // 2, 128, 256, 2), (ng, S // 4, 256, 512, 2), (ng, S // 8, 512, 512, 2),
          (nd, S // 2, 64, 128, 2), (nd, S // 4, 128, 256, 2), (nd, S // 8, 256, 512, 2), (n1, S, 64, 128, 2)]
if args:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in args]
BF = torch.bfloat16
for n, h, cin, cout, stride in SHAPES:
    ho = h // stride
    x = torch.randn((n, h, h, cin), device="cuda").to(BF)
    dy = torch.randn((n, ho, ho, cout), device="cuda").to(BF)
    dw = torch.empty((3, 3, cin, cout), device="cuda")
    if blocks or ab:
        ops.set_tuning("wgrad.blocks", blocks or max(ab))
    ws = torch.empty(ops.conv2d_wgrad_workspace(n, ho, ho, cin, cout, 3) // 4 + 1024, device="cuda")
    fn = lambda: ops.conv2d_wgrad(x, None, 0, cin, 0, dy, cout, dw, n, h, h, cin, cin, cout, 3, stride, 0, ws)
    flops = 2.0 * n * ho * ho * 9 * cin * cout
    times, names, res = {0: [], 1: []}, {}, {}
    for _ in range(5):
        for wide in (0, 1):
            if ab:
                ops.set_tuning("wgrad.blocks", ab[wide])
            else:
                ops.set_tuning("wgrad.bf16_wide", 1 if wide else 4)
            fn()
            names[wide] = ops.last_kernel()
            res[wide] = dw.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[wide].append(e0.elapsed_time(e1) / 4 * 1e3)
    ops.set_tuning("reset", 0)
    err = float((res[0] - res[1]).norm() / res[1].norm())
    t0, t1 = statistics.median(times[0]), statistics.median(times[1])
    if ab:
        names = {i: f"{names[i]} @{ab[i]}" for i in (0, 1)}
    print(f"n{n} h{h} {cin}x{cout} s{stride}: {names[0]} {t0:7.1f} us ({flops / t0 / 1e6:6.1f} TF) | {names[1]} {t1:7.1f} us ({flops / t1 / 1e6:6.1f} TF) | "
          f"{100 * (t0 / t1 - 1):+5.1f} %  (rel diff {err:.1e})", flush=True)
