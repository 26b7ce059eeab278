"""A/B of the consumer-side InstanceNorm (shm_conv2d_in_fwd_norm / shm_conv2d_wgrad_norm) per layer shape: the folding kernel on
the un-normalised tensor against the plain kernel on the normalised one, and the stand-alone pass (shm_in_apply) the fold removes.
usage: bench_fold.py [f32|bf16|both] [exact|scaled] [n,h,cin,cout[,c1] ...]   (c1 > 0: Concatenate, the second source is the folded one;
exact = SHM_NORM_EXACT, the in-LDS normalisation; scaled = SHM_NORM_SCALED, per-sample weights / bias rows, whose preparation kernel and
rank-n weight-gradient term are timed beside the kernels)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from shmgan_amd import ops

args = sys.argv[1:]
which = "both"
if args and args[0] in ("f32", "bf16", "both"):
    which, args = args[0], args[1:]
mode = "exact"
if args and args[0] in ("exact", "scaled"):
    mode, args = args[0], args[1:]
MODE = ops.NORM_SCALED if mode == "scaled" else ops.NORM_EXACT
SHAPES = [(40, 256, 64, 64, 0), (40, 256, 128, 64, 64), (40, 128, 128, 128, 0), (40, 128, 256, 128, 128), (40, 64, 256, 256, 0), (8, 256, 64, 64, 0)]
if args:
    SHAPES = [tuple(int(v) for v in a.split(",")) for a in args]
    SHAPES = [s if len(s) == 5 else s + (0,) for s in SHAPES]


def timeit(fn, reps=6):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dt in [d for d, k in ((torch.float32, "f32"), (torch.bfloat16, "bf16")) if which in (k, "both")]:
    for n, h, cin, cout, c1 in SHAPES:
        cs = cin - c1                                  # channels of the folded source
        u = torch.randn((n, h, h, c1), device="cuda").to(dt) if c1 else None
        a = (torch.randn((n, h, h, cs), device="cuda") * 1.5 + 0.3).to(dt)
        beta = torch.zeros(cs, device="cuda")
        st = torch.zeros(n * cs * 2, dtype=torch.float64, device="cuda")
        ops.in_stats(a, cs, st, n, h * h, cs, 1e-6)
        nt = torch.empty((n, 4, cs), device="cuda")
        ops.in_norm_table(st, beta, nt, n, cs)
        ahat = torch.empty_like(a)
        dy = torch.randn((n, h, h, cout), device="cuda").to(dt)
        wk = (torch.randn(9 * cout * cin, device="cuda") * 0.05).to(dt)
        y = torch.empty((n, h, h, cout), device="cuda", dtype=dt)
        dw = torch.empty((3, 3, cin, cout), device="cuda")
        stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
        scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
        ws = torch.empty(ops.conv2d_wgrad_norm_workspace(n, h, h, cin, cout, 3, dt) // 4 + 1024, device="cuda")
        wk_n = torch.empty((n, 9 * cout * cin), device="cuda", dtype=dt)
        bias_n = torch.empty((n, cout), device="cuda")
        bias0 = torch.zeros(cout, device="cuda")
        dzsum = torch.zeros((n, cout), dtype=torch.float64, device="cuda")
        flops = 2.0 * n * h * h * 9 * cin * cout
        x1, x2 = (u, None) if c1 else (None, None)

        def prep():
            ops.conv2d_norm_prepare(wk, bias0, nt, cs, c1, wk_n, bias_n, n, cin, cout, 3)

        def fwd(src, **kw):
            folded = bool(kw)
            w_, b_ = (wk_n, bias_n) if (folded and MODE) else (wk, bias0)
            if folded:
                kw["norm_mode"] = MODE
            if c1:
                ops.conv2d_in_fwd(u, src, c1, c1, cs, w_, b_, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr, **kw)
            else:
                ops.conv2d_in_fwd(src, None, 0, cs, 0, w_, b_, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr, **kw)

        def wgrad(src, **kw):
            folded = bool(kw)
            if folded:
                kw["norm_mode"] = MODE
            if c1:
                ops.conv2d_wgrad(u, src, c1, c1, cs, dy, cout, dw, n, h, h, cin, cin, cout, 3, 1, 0, ws, **kw)
            else:
                ops.conv2d_wgrad(src, None, 0, cs, 0, dy, cout, dw, n, h, h, cin, cin, cout, 3, 1, 0, ws, **kw)
            if folded and MODE:
                ops.conv2d_wgrad_norm_finish(dw, nt, dzsum, n, cs, c1, cin, cout, 3)
        key = "nt_x2" if c1 else "nt_x"
        t_apply = timeit(lambda: ops.in_apply(a, cs, st, beta, ahat, cs, n, h * h, cs))
        t_prep = timeit(prep) if MODE else 0.0
        t_f0, t_f1 = timeit(lambda: fwd(ahat)), timeit(lambda: fwd(a, **{key: nt}))
        kf = ops.last_kernel()
        t_w0, t_w1 = timeit(lambda: wgrad(ahat)), timeit(lambda: wgrad(a, **{key: nt}))
        kw_ = ops.last_kernel()
        print(f"{str(dt)[6:]:9s} n{n} h{h} {cin}->{cout} c1={c1}: in_apply {t_apply:7.1f} us | fwd {t_f0:7.1f} -> {t_f1:7.1f} us ({flops / t_f1 / 1e6:6.1f} TF, "
              f"{100 * (t_f1 / t_f0 - 1):+5.1f} %) {kf.split('<')[0]} | wgrad {t_w0:7.1f} -> {t_w1:7.1f} us ({100 * (t_w1 / t_w0 - 1):+5.1f} %) {kw_.split('<')[0]}"
              f" | prepare {t_prep:5.1f} us | fold saves {t_apply - t_prep - (t_f1 - t_f0) - (t_w1 - t_w0):7.1f} us", flush=True)
