"""Inference throughput of the evaluation path (/root/reference/test.py:218-297 = trainer.infer): standardised
YUV -> SpecSeg mask -> G once -> five cyclic G calls per input image.  python tools/bench_infer.py [float32|bfloat16] [B]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from shmgan_amd import ShmGANwithSSpecSeg

dt = sys.argv[1] if len(sys.argv) > 1 else "float32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = 256
m = ShmGANwithSSpecSeg(image_size=S, filter_size=64, batch_size=B, compute_dtype=dt).build()
x = torch.from_numpy(np.random.default_rng(0).random((B, S, S, 3), dtype=np.float32)).cuda()
for _ in range(3):
    m.infer(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    m.infer(x)
torch.cuda.synchronize()
dt_s = (time.perf_counter() - t0) / n
print(f"{dt} B={B}: {dt_s * 1e3:.2f} ms per batch = {B / dt_s:.1f} input images/s ({6 * B / dt_s:.0f} generator forwards/s)")
