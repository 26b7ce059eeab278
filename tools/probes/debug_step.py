"""Diagnostic: per-stage error of one train_step against the float64 oracle (GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
import numpy as np, torch
from oracle import step_torch as st
from util import rel_l2, host, cosine
from shmgan_amd import ShmGANwithSSpecSeg

S, F, B, step = 64, 16, 1, 0
if len(sys.argv) > 1: S, F, B, step = map(int, sys.argv[1:5])
m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
g, d, gb, db = st.init_params(F, S)
inp = st.make_inputs(B, S); dr = st.make_draws(step, B, S, F); sf = st.style_factor_intended(S)
ref = st.train_step(g, d, gb, db, inp, dr, sf, F)
m.train_step(*inp, draws=dr, style_factor=sf, apply=False)
torch.cuda.synchronize()
A = m.arena
print("flags", dr.flags)
print("gen_Y", rel_l2(host(m.gen_Y), ref["outs"]["gen_Y"].numpy()))
dcyc = host(A.get("loss/dcyc_y", (5 * B, S, S, 1)))
dgen = host(A.get("loss/dgen_y", (B, S, S, 1)))
rc = np.concatenate([t.numpy() for t in ref["dcyc_Y"]], 0)
print("dcyc_y total", rel_l2(dcyc, rc), [rel_l2(dcyc[k*B:(k+1)*B], rc[k*B:(k+1)*B]) for k in range(5)])
print("dgen_y total", rel_l2(dgen, ref["dgen_Y"].numpy()), cosine(dgen, ref["dgen_Y"].numpy()))
for name, P, rg in (("D", m.D.P, ref["gD"]), ("G", m.G.P, ref["gG"])):
    print(name, " ".join(f"{rel_l2(host(a), b.numpy()):.1e}" for a, b in zip(P.grads, rg)))
