"""Does a result depend on what the allocator's blocks held before?  Runs the same step on fresh memory and after filling 20 GB
with a recognisable value (the caching allocator then hands those blocks to the next model's arena) and reports which outputs /
gradient tensors differ.  python tools/probes/poison_probe.py [S] [dtype]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from oracle import step_torch as st
from shmgan_amd import ShmGANwithSSpecSeg
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dts = sys.argv[2:] or ["bfloat16", "float32"]
F, B = 64, 1
inp, dr = st.make_inputs(B, S), st.make_draws(6, B, S, F)
def run(dt):
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype=dt).build()
    m.train_step(*inp, draws=dr, apply=False)
    torch.cuda.synchronize()
    out = dict(l=m.losses(), gy=m.gen_Y.clone().cpu(), rf=m.D.ctx["rf"].clone().cpu(), cls=m.D.ctx["cls"].clone().cpu(),
               cyc=m.cyc_genED_rgb.clone().cpu(), cyc0=m.cyc_gen0_rgb.clone().cpu(),
               gg=[g.clone().cpu() for g in m.G.P.grads], dg=[g.clone().cpu() for g in m.D.P.grads],
               arena={str(k): v.clone().cpu() for k, v in m.arena.t.items() if v.dtype != torch.float64 and v.numel() < 400e6})
    del m
    torch.cuda.empty_cache()
    return out
def poison(val):
    t = torch.full((int(20e9) // 4,), val, device="cuda")
    torch.cuda.synchronize()
    del t
for dt in dts:
    torch.cuda.empty_cache()
    a = run(dt)
    poison(-7.0e-3)
    b = run(dt)
    md = lambda x, y: float((x.float() - y.float()).abs().max())
    print(dt, "gen_Y", md(a["gy"], b["gy"]), "cyc0", md(a["cyc0"], b["cyc0"]), "cycED", md(a["cyc"], b["cyc"]), "rf rows", [round(md(a["rf"][i], b["rf"][i]), 6) for i in range(12 * B)],
          "cls rows", [round(md(a["cls"][i], b["cls"][i]), 6) for i in range(12 * B)], flush=True)
    print(dt, "G grads differing:", [(i, round(md(x, y), 6)) for i, (x, y) in enumerate(zip(a["gg"], b["gg"])) if md(x, y) > 0][:60], flush=True)
    print(dt, "D grads differing:", [(i, round(md(x, y), 6)) for i, (x, y) in enumerate(zip(a["dg"], b["dg"])) if md(x, y) > 0], flush=True)
    bad = []
    for k, x in a["arena"].items():
        y = b["arena"].get(k)
        if y is not None and x.shape == y.shape:
            d = md(x, y)
            if d > 0 or not bool(torch.isfinite(x.float()).all()):
                bad.append((k, round(d, 6)))
    print(dt, "arena buffers differing:", bad, flush=True)
    print(dt, "losses differing:", {k: (v, b["l"][k]) for k, v in a["l"].items() if k != "ssim" and v != b["l"][k]}, flush=True)
