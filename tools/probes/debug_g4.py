import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
import numpy as np, torch
import torch.nn.functional as Fn
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
x16 = A.get("g1/in", (B, S, S, 16)).clone()
rng = np.random.default_rng(5)
dy = dev(rng.standard_normal((B, S, S, 1)))
gv = [t64(a).requires_grad_(True) for a in g]
xt = t64(host(x16)[..., :10]).requires_grad_(True)
rec = []
yt = st.generator_forward(gv, [t64(b) for b in gb], xt, F, record=rec)
yt.backward(t64(host(dy)))
m.G.debug = {}
m.G.forward(x16, "t"); m.G.zero_grad(); m.G.backward(dy, "t", need_dx=True); m.G.finish_grads(); torch.cuda.synchronize()
c = m.G.ctx["t"]
# record index -> layer index
lis = [r["li"] for r in c["recs"]]
for ri, li in enumerate(lis):
    z, ah = rec[ri]
    g1, g2, dz = m.G.debug[li]
    dah = ah.grad.permute(0, 2, 3, 1).numpy()     # total gradient at the IN output
    mine = host(g1)
    if g2 is not None:
        mine = mine + 0.25 * np.repeat(np.repeat(host(g2), 2, axis=1), 2, axis=2)
    a_ref = Fn.leaky_relu(z.detach(), 0.2).permute(0, 2, 3, 1).numpy()
    print(f"li {li:2d} h {z.shape[2]:3d} c {z.shape[1]:4d}  a {rel_l2(host(c['recs'][ri]['a']), a_ref):.1e}  d_ahat {rel_l2(mine, dah):.1e}  dz {rel_l2(host(dz), z.grad.permute(0,2,3,1).numpy()):.1e}")
ri = lis.index(14)
z, ah = rec[ri]
g1, g2, dz = m.G.debug[14]
np.savez("gpurun_out/l14.npz", a=host(c['recs'][ri]['a']), stats=c['recs'][ri]['stats'].cpu().numpy(), g1=host(g1), dz=host(dz),
         z=z.detach().permute(0,2,3,1).numpy(), dz_ref=z.grad.permute(0,2,3,1).numpy(), dah_ref=ah.grad.permute(0,2,3,1).numpy())
