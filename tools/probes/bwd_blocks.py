"""Block-count sweeps of the bf16 InstanceNorm-backward passes (reduce: "elem.reduce_blocks", apply: "elem.apply_blocks") on the step's
layer shapes: us for the reduce + apply pair (shm_in_bwd) per setting.  usage: bwd_blocks.py [reduce|apply]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from shmgan_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "reduce"
key = "elem.reduce_blocks" if which == "reduce" else "elem.apply_blocks"
vals = [0, 256, 384, 512, 768, 1024, 1536, 2048, 4096] if which == "reduce" else [1024, 2048, 4096, 8192, 16384]
dt = torch.bfloat16
for n, h, c in [(40, 256, 64), (40, 128, 128), (40, 64, 256), (40, 32, 512), (8, 256, 64), (96, 128, 64)]:
    a = torch.randn((n, h, h, c), device="cuda").to(dt)
    g = torch.randn((n, h, h, c), device="cuda").to(dt)
    out = torch.empty_like(a)
    stats = torch.empty(n * c * 2, dtype=torch.float64, device="cuda")
    red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
    db = torch.zeros(c, dtype=torch.float64, device="cuda")
    ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
    fn = lambda: ops.in_bwd(g, c, None, 0, a, c, stats, red, out, c, db, n, h, h, c, 0.2)
    res = []
    for v in vals:
        ops.set_tuning(key, v)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 8 * 1e3)
    ops.set_tuning("reset", 0)
    nbytes = 5 * 2 * n * h * h * c
    print(f"n{n} h{h} c{c} {key}: " + "  ".join(f"{v}:{t:6.1f}us" for v, t in zip(vals, res)) + f"   best {nbytes / min(res) / 1e3:5.0f} GB/s", flush=True)
