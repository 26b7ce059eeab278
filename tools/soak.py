"""Full-size soak: N optimizer steps at S=256, B=8 on fixed synthetic data; prints the loss trajectory and
checks that nothing goes non-finite.  python tools/soak.py [float32|bfloat16] [steps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from shmgan_amd import ShmGANwithSSpecSeg

dt = sys.argv[1] if len(sys.argv) > 1 else "float32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B, S = 8, 256
m = ShmGANwithSSpecSeg(image_size=S, filter_size=64, batch_size=B, compute_dtype=dt, g_lr=2e-4).build()
rng = np.random.default_rng(0)
base = rng.random((B, S, S, 3), dtype=np.float32)
inp = [torch.from_numpy(np.clip(base + 0.1 * rng.standard_normal(base.shape).astype(np.float32), 0, 1)).cuda() for _ in range(5)]
rows = []
for it in range(steps):
    m.train_step(*inp)
    if it % 10 == 0 or it == steps - 1:
        l = m.losses()
        assert all(np.isfinite(v) for k, v in l.items() if k != "ssim"), (it, l)
        rows.append((it, l["total_Generator_loss"], l["total_Discriminator_loss"], l["L1_loss_Gen"], l["ssim_cyc_loss"]))
        print(f"{dt} step {it:4d}  G {rows[-1][1]:10.4f}  D {rows[-1][2]:10.4f}  L1 {rows[-1][3]:8.4f}  ssim {rows[-1][4]:8.5f}", flush=True)
w = torch.cat([m.G.P.flat, m.D.P.flat])
assert torch.isfinite(w).all()
print("ok: weights finite, L1", rows[0][3], "->", rows[-1][3])
