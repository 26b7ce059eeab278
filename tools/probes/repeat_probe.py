"""Run-to-run reproducibility of one train_step (apply=False) on fixed inputs: repeats the step and reports every gradient
tensor / named loss whose value differs from the first repetition by more than the f64-atomics bound.
python tools/probes/repeat_probe.py [S] [B] [dtype] [reps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from oracle import step_torch as st
from shmgan_amd import ShmGANwithSSpecSeg
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dt = sys.argv[3] if len(sys.argv) > 3 else "bfloat16"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
F = 64
inp, dr = st.make_inputs(B, S), st.make_draws(6, B, S, F)
m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B, compute_dtype=dt).build()
ref = None
for r in range(reps):
    m.train_step(*inp, draws=dr, apply=False)
    torch.cuda.synchronize()
    cur = dict(l=dict(m.losses()), gy=m.gen_Y.clone(), rf=m.D.ctx["rf"].clone(), cls=m.D.ctx["cls"].clone(),
               gg=[g.clone() for g in m.G.P.grads], dg=[g.clone() for g in m.D.P.grads])
    if ref is None:
        ref = cur
        continue
    md = lambda x, y: float((x.float() - y.float()).abs().max())
    gg = [(i, md(x, y), float(x.abs().max())) for i, (x, y) in enumerate(zip(ref["gg"], cur["gg"])) if md(x, y) > 0]
    dg = [(i, md(x, y), float(x.abs().max())) for i, (x, y) in enumerate(zip(ref["dg"], cur["dg"])) if md(x, y) > 0]
    ls = {k: (v, cur["l"][k]) for k, v in ref["l"].items() if k != "ssim" and abs(v - cur["l"][k]) > 1e-9 * max(1.0, abs(v))}
    print(f"{dt} S{S} B{B} rep {r}: gen_Y {md(ref['gy'], cur['gy'])} rf {md(ref['rf'], cur['rf'])} cls {md(ref['cls'], cur['cls'])} "
          f"G grads differing {len(gg)} (worst {max(gg, key=lambda t: t[1] / max(t[2], 1e-30)) if gg else None}) "
          f"D grads differing {len(dg)} (worst {max(dg, key=lambda t: t[1] / max(t[2], 1e-30)) if dg else None}) losses {ls}", flush=True)
