"""Does running the cyclic generator forward as two half-batches on two streams beat one full-batch pass
(the halves fill each other's launch tails)?  Timing probe only."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from shmgan_amd import ShmGANwithSSpecSeg

dt = sys.argv[1] if len(sys.argv) > 1 else "float32"
B, S = 8, 256
m = ShmGANwithSSpecSeg(image_size=S, filter_size=64, batch_size=B, compute_dtype=dt).build()
G = m.G
x = torch.randn((5 * B, S, S, G.pad), device="cuda").to(m.compute_dtype)
G.prepare_weights()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def full():
    G.forward(x, "p_full")


def split(k=2):
    n = 5 * B
    ev = torch.cuda.Event()
    ev.record()
    streams = [s1, s2][:k]
    step = n // k
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            st.wait_event(ev)
            G.forward(x[i * step:(i + 1) * step].contiguous() if False else x[i * step:(i + 1) * step], f"p_half{i}")
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)


def timeit(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print(dt, "full n=40:", round(timeit(full), 2), "ms;  two halves on two streams:", round(timeit(split), 2), "ms")
