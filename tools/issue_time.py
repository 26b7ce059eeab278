"""Host-side issue time of one train_step (Python + ctypes + HIP launch calls, no synchronisation) against its
GPU time: tells whether a configuration is launch-bound."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from shmgan_amd import ShmGANwithSSpecSeg

dt = sys.argv[1] if len(sys.argv) > 1 else "float32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = ShmGANwithSSpecSeg(image_size=S, filter_size=64, batch_size=B, compute_dtype=dt).build()
rng = np.random.default_rng(0)
inp = [torch.from_numpy(rng.random((B, S, S, 3), dtype=np.float32)).cuda() for _ in range(5)]
for _ in range(3):
    m.train_step(*inp)
torch.cuda.synchronize()
issue, total = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_step(*inp)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print(f"{dt} B={B} S={S}: host issue {np.median(issue):.2f} ms, step {np.median(total):.2f} ms")
