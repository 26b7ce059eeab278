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
x_real = A.get("g1/in", (B, S, S, 16)).clone()
dy_real = A.get("loss/dgen_y", (B, S, S, 1)).clone()
rng = np.random.default_rng(5)
x_rand = torch.zeros_like(x_real); x_rand[..., :10] = dev(rng.standard_normal((B, S, S, 10)))
dy_rand = dev(rng.standard_normal((B, S, S, 1)))
def run(x16, dy, label):
    gv = [t64(a).requires_grad_(True) for a in g]
    xt = t64(host(x16)[..., :10]).requires_grad_(True)
    yt = st.generator_forward(gv, [t64(b) for b in gb], xt, F)
    grads = torch.autograd.grad(yt, gv + [xt], t64(host(dy)))
    # fp32 oracle for the noise floor
    gv32 = [torch.tensor(a, dtype=torch.float32, requires_grad=True) for a in g]
    yt32 = st.generator_forward(gv32, [torch.tensor(b, dtype=torch.float32) for b in gb], torch.tensor(host(x16)[..., :10], dtype=torch.float32), F)
    g32 = torch.autograd.grad(yt32, gv32, torch.tensor(host(dy), dtype=torch.float32))
    m.G.forward(x16, "t"); m.G.zero_grad(); m.G.backward(dy, "t", need_dx=True); m.G.finish_grads(); torch.cuda.synchronize()
    print(label, "hip :", " ".join(f"{rel_l2(host(a), b.numpy()):.0e}" for a, b in zip(m.G.P.grads[::2], grads[:-1:2])))
    print(label, "fp32:", " ".join(f"{rel_l2(a.double().numpy(), b.numpy()):.0e}" for a, b in zip(g32[::2], grads[:-1:2])))
    c = m.G.ctx["t"]
    print(label, "max inv per IN layer:", " ".join(f"{float(r['stats'].view(-1,2)[:,1].max()):.0f}" for r in c["recs"]))
run(x_real, dy_real, "real/real")
run(x_real, dy_rand, "real/rand")
run(x_rand, dy_real, "rand/real")
