"""Diagnostic: per-tensor error of a generator forward/backward against the float64 oracle."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
from oracle import step_torch as st
from util import rel_l2, host, cosine, t64, dev
from shmgan_amd import ShmGANwithSSpecSeg

S, F, B = 64, 16, 5
if len(sys.argv) > 1: S, F, B = map(int, sys.argv[1:4])
m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
g, d, gb, db = st.init_params(F, S)
rng = np.random.default_rng(30)
x = rng.standard_normal((B, S, S, 10)); dy = rng.standard_normal((B, S, S, 1))
gv = [t64(a).requires_grad_(True) for a in g]
xt = t64(x).requires_grad_(True)
yt = st.generator_forward(gv, [t64(b) for b in gb], xt, F)
grads = torch.autograd.grad(yt, gv + [xt], t64(dy))
x16 = torch.zeros((B, S, S, 16), device="cuda"); x16[..., :10] = dev(x)
y = m.G.forward(x16, "t")
print("y", rel_l2(host(y), yt.detach().numpy()))
m.G.zero_grad()
dx = m.G.backward(dev(dy), "t", need_dx=True)
m.G.finish_grads(); torch.cuda.synchronize()
print("dx", rel_l2(host(dx)[..., :10], grads[-1].numpy()))
print(" ".join(f"{i}:{rel_l2(host(a), b.numpy()):.1e}" for i, (a, b) in enumerate(zip(m.G.P.grads, grads[:-1]))))
