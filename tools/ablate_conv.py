"""Timing-only ablations of the tap-GEMM kernel (outputs are WRONG in the ablated builds)."""
import ctypes as C, subprocess, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "shmgan_amd" / "csrc"
variants = {"base": [], "nodma": ["-DSHM_ABL_NODMA"], "fixaddr": ["-DSHM_ABL_FIXADDR"], "pin": ["-DSHM_SCHED_PIN"], "sameline": ["-DSHM_ABL_SAMELINE"], "noaddr": ["-DSHM_ABL_NOADDR"], "nobar": ["-DSHM_ABL_NOBAR"], "noload": ["-DSHM_ABL_NOLOAD"], "nostore": ["-DSHM_ABL_NOSTORE"]
            }
import os
if os.environ.get("ABL_ONLY"): variants = {k: v for k, v in variants.items() if k in os.environ["ABL_ONLY"].split(",")}
shapes = [(40, 64, 256, 256), (40, 128, 128, 128), (40, 256, 64, 64), (40, 32, 512, 512)]
for name, flags in variants.items():
    so = f"/tmp/abl_{name}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-munsafe-fp-atomics", *flags,
                    str(src / "conv_igemm.hip"), str(src / "norm_elem.hip"), "-o", so], check=True)
    L = C.CDLL(so)
    res = []
    for n, h, cin, cout in shapes:
        x = torch.randn(n, h, h, cin, device="cuda"); w = torch.randn(9 * cout * cin, device="cuda") * 0.05
        y = torch.empty(n, h, h, cout, device="cuda")
        P = C.c_void_p
        def run():
            L.shm_conv2d_fwd(P(x.data_ptr()), None, 0, cin, 0, P(w.data_ptr()), None, P(y.data_ptr()), cout, n, h, h, cin, cout, 3, 1, C.c_float(0.2), None)
        for _ in range(3): run()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(2.0 * n * h * h * 9 * cin * cout / ms / 1e9)
    print(f"{name:16s}", " ".join(f"{r:7.1f}" for r in res), "TFLOP/s", flush=True)
