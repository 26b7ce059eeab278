"""Where does a train_step spend its time?  Event markers between the sections of the step."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from shmgan_amd import ShmGANwithSSpecSeg, ops
import shmgan_amd.trainer as T

S, F, B = 256, 64, 8
m = ShmGANwithSSpecSeg(image_size=S, filter_size=F, batch_size=B).build()
rng = np.random.default_rng(0)
inp = [torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).cuda() for _ in range(5)]
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
# monkeypatch section boundaries through the model methods
G, D = m.G, m.D
def wrap(obj, meth, label):
    f = getattr(obj, meth)
    def g(*a, **k):
        mark(label + ":begin"); r = f(*a, **k); mark(label + ":end"); return r
    setattr(obj, meth, g)
wrap(G, "forward", "G.forward"); wrap(G, "backward", "G.backward"); wrap(D, "forward", "D.forward")
wrap(D, "backward_params", "D.bwd_params"); wrap(D, "backward_input", "D.bwd_input")
for mode in ("lane", "serial"):
    if mode == "serial": m._get_lane().stream = None
    for it in range(3):
        marks.clear(); mark("step:begin"); m.train_step(*inp); mark("step:end"); torch.cuda.synchronize()
    t0 = marks[0][1]
    print("mode", mode)
    prev = None
    for name, e in marks:
        t = t0.elapsed_time(e)
        print(f"  {name:22s} {t:8.2f} ms" + (f"   (+{t - prev:6.2f})" if prev is not None else ""))
        prev = t
