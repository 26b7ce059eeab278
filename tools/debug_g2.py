"""Diagnostic: generator backward on the REAL cyclic inputs / loss gradient of a step."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from oracle import step_torch as st
from util import rel_l2, host, cosine, t64, dev
from shmgan_amd import ShmGANwithSSpecSeg

S, F, B, step = 64, 16, 1, 0
m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
g, d, gb, db = st.init_params(F, S)
inp = st.make_inputs(B, S); dr = st.make_draws(step, B, S, F); sf = st.style_factor_intended(S)
m.train_step(*inp, draws=dr, style_factor=sf, apply=False)
torch.cuda.synchronize()
A = m.arena
for tag, n, dyname in (("cyc", 5 * B, "loss/dcyc_y"), ("g1", B, "loss/dgen_y")):
    x16 = A.get(f"{tag}/in", (n, S, S, 16)).clone()
    dy = A.get(dyname, (n, S, S, 1)).clone()
    gv = [t64(a).requires_grad_(True) for a in g]
    xt = t64(host(x16)[..., :10]).requires_grad_(True)
    yt = st.generator_forward(gv, [t64(b) for b in gb], xt, F)
    grads = torch.autograd.grad(yt, gv + [xt], t64(host(dy)))
    m.G.zero_grad()
    dx = m.G.backward(dy, tag, need_dx=True)
    m.G.finish_grads(); torch.cuda.synchronize()
    print(tag, "dy mean/std", float(dy.mean()), float(dy.std()), "dx", rel_l2(host(dx)[..., :10], grads[-1].numpy()))
    print(" ".join(f"{i}:{rel_l2(host(a), b.numpy()):.1e}" for i, (a, b) in enumerate(zip(m.G.P.grads, grads[:-1]))))
    # same thing on a fresh forward (tag "t")
    y = m.G.forward(x16, "t")
    m.G.zero_grad()
    dx = m.G.backward(dy, "t", need_dx=True)
    m.G.finish_grads(); torch.cuda.synchronize()
    print(tag, "fresh fwd:", " ".join(f"{i}:{rel_l2(host(a), b.numpy()):.1e}" for i, (a, b) in enumerate(zip(m.G.P.grads, grads[:-1]))))
