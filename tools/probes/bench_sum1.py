"""First-layer input-gradient stencil (shm_conv3x3_dgrad_sum1) on the step's three launches, us and GB/s of the dz tensor read."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from shmgan_amd import ops

for dt in (torch.bfloat16, torch.float32):
    for nk, batch, h, c, stride in [(5, 8, 256, 64, 1), (1, 40, 256, 64, 2), (1, 8, 256, 64, 2)]:
        ho = h // stride
        dz = torch.randn((nk * batch, ho, ho, c), device="cuda").to(dt)
        weff = torch.randn((nk, 9, c), device="cuda")
        out = torch.zeros((batch, h, h, 1), device="cuda")
        fn = lambda: ops.conv3x3_dgrad_sum1(dz, c, weff, out, nk, batch, h, h, c, stride, 0)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{str(dt):15s} nk{nk} n{batch} h{h} c{c} s{stride}: {us:7.1f} us  {dz.numel() * dz.element_size() / us / 1e3:6.0f} GB/s", flush=True)
