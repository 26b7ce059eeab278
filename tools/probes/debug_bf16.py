"""Per-tensor gradient agreement of the bf16 step with the float64 oracle (diagnostics)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
import numpy as np
import torch
from oracle import step_torch as st
from util import cosine, host, rel_l2
from shmgan_amd import ShmGANwithSSpecSeg

S, F_, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g, d, gb, db = st.init_params(F_, S)
inp = st.make_inputs(B, S)
dr = st.make_draws(0, B, S, F_)
sf = st.style_factor_intended(S)
ref = st.train_step(g, d, gb, db, inp, dr, sf, F_)
for dt, gdt in (("float32", "float32"), ("bfloat16", "float32"), ("bfloat16", "bfloat16")):
    m = ShmGANwithSSpecSeg(image_size=S, filter_size=F_, batch_size=B, compute_dtype=dt, grad_dtype=gdt).build()
    m.G.set_weights(g); m.D.set_weights(d); m.G.set_betas(gb); m.D.set_betas(db)
    m.train_step(*inp, draws=dr, style_factor=sf, apply=False)
    torch.cuda.synchronize()
    got = m.losses()
    print(dt, gdt, {k: (round(got[k], 5), round(v, 5)) for k, v in ref["losses"].items()})
    print(" gen_Y rel", rel_l2(host(m.gen_Y), ref["outs"]["gen_Y"].numpy()))
    for name, P, rg in (("D", m.D.P, ref["gD"]), ("G", m.G.P, ref["gG"])):
        cs = [round(cosine(host(a), r.numpy()), 4) for a, r in zip(P.grads, rg) if float(r.norm()) > 1e-12]
        print(" ", name, cs)
    if dt == "bfloat16":       # same comparison with the oracle taking the device's side of every LeakyReLU kink
        masks = {"g1": m.G.lrelu_masks("g1"), "cyc": m.G.lrelu_masks("cyc"), "d": m.D.lrelu_masks()}
        pin = st.train_step(g, d, gb, db, inp, dr, sf, F_, masks=masks)
        for name, P, rg in (("D", m.D.P, pin["gD"]), ("G", m.G.P, pin["gG"])):
            cs = [round(cosine(host(a), r.numpy()), 4) for a, r in zip(P.grads, rg) if float(r.norm()) > 1e-12]
            print("  pinned", name, cs)
