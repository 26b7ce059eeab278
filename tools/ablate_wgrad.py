"""Timing-only ablations of the weight-gradient kernel (outputs are WRONG in the ablated builds)."""
import ctypes as C, subprocess, sys, os
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "shmgan_amd" / "csrc"
variants = {"base": [], "sameline": ["-DSHM_ABL_SAMELINE"], "noaddr": ["-DSHM_ABL_NOADDR"], "nobar": ["-DSHM_ABL_NOBAR"], "noload": ["-DSHM_ABL_NOLOAD"], "nostore": ["-DSHM_ABL_NOSTORE"],
            "noload_nostore": ["-DSHM_ABL_NOLOAD", "-DSHM_ABL_NOSTORE"]}
if os.environ.get("ABL_ONLY"): variants = {k: v for k, v in variants.items() if k in os.environ["ABL_ONLY"].split(",")}
shapes = [(40, 256, 64, 64), (40, 128, 128, 128), (40, 64, 256, 256), (40, 32, 512, 512)]
for name, flags in variants.items():
    so = f"/tmp/ablw_{name}.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-munsafe-fp-atomics", *flags,
                    str(src / "conv_wgrad.hip"), str(src / "norm_elem.hip"), "-o", so], check=True)
    L = C.CDLL(so)
    L.shm_conv2d_wgrad_workspace.restype = C.c_size_t
    res = []
    for n, h, cin, cout in shapes:
        x = torch.randn(n, h, h, cin, device="cuda"); dy = torch.randn(n, h, h, cout, device="cuda")
        dw = torch.empty(9 * cin * cout, device="cuda")
        wsb = L.shm_conv2d_wgrad_workspace(n, h, h, cin, cout, 3)
        ws = torch.empty(wsb // 4 + 16, device="cuda")
        P = C.c_void_p
        def run():
            rc = L.shm_conv2d_wgrad(P(x.data_ptr()), None, 0, cin, 0, P(dy.data_ptr()), cout, P(dw.data_ptr()), n, h, h, cin, cin, cout, 3, 1, 0,
                                    P(ws.data_ptr()), C.c_size_t(wsb), None)
            assert rc == 0
        for _ in range(3): run()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(2.0 * n * h * h * 9 * cin * cout / ms / 1e9)
    print(f"{name:16s}", " ".join(f"{r:7.1f}" for r in res), "TFLOP/s (incl. reduce)", flush=True)
