"""Phase stamps of in_bwd_fused8_kernel (timing-only build of norm_elem.hip with -DSHM_FUSED_STAMP, loaded through SHM_LIB_PATH): per block
start, slices loaded, rows written, arrival counted, released, phase 2 stores issued, departure counted, end -- microseconds from the first start.

    python tools/probes/in_bwd_fused_stamps.py [n,h,c]"""
import os
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
so = ROOT / "build_ab" / "libshm_fused_stamp.so"
if "SHM_LIB_PATH" not in os.environ:
    from shmgan_amd import _lib
    if not so.exists() or so.stat().st_mtime < (_lib.CSRC / "norm_elem.hip").stat().st_mtime:
        so.parent.mkdir(exist_ok=True)
        _lib.build()
        obj = so.parent / (so.stem + ".o")
        flags = [f for f in _lib.HIPCC_FLAGS if f != "-shared"]
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-DSHM_FUSED_STAMP", "-c", str(_lib.CSRC / "norm_elem.hip"), "-o", str(obj)], check=True)
        objs = [str(obj) if s == "norm_elem.hip" else str(_lib.CSRC / "_obj" / (Path(s).stem + ".o")) for s in _lib.SOURCES]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", str(so)], check=True)
    if "--build-only" in sys.argv:
        sys.exit(0)
    os.environ["SHM_LIB_PATH"] = str(so)
    sys.exit(subprocess.run([sys.executable, *sys.argv], env=os.environ).returncode)
import numpy as np
import torch
from shmgan_amd import ops

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, h, c = (int(v) for v in (args[0] if args else "40,256,64").split(","))
BF = torch.bfloat16
a = torch.randn(n, h, h, c, device="cuda").to(BF)
g = torch.randn(n, h, h, c, device="cuda").to(BF)
stats = torch.zeros(n * c * 2, dtype=torch.float64, device="cuda")
ops.in_stats(a, c, stats, n, h * h, c, 1e-6)
dz = torch.empty_like(a)
db = torch.zeros(c, dtype=torch.float64, device="cuda")
red = torch.zeros(n * c * 3, dtype=torch.float64, device="cuda")
bpi = h * h * c // 16384
base = ops.in_bwd_fused_doubles(n, h * h, c)
scratch = torch.zeros(base + n * bpi * 8 + 8, dtype=torch.float64, device="cuda")
for _ in range(4):
    ops.in_bwd(g, c, None, 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=scratch)
    assert "fused" in ops.last_kernel()
torch.cuda.synchronize()
st = scratch[base:base + n * bpi * 8].view(torch.int64).cpu().numpy().reshape(n, bpi, 8).astype(np.float64)
st = (st - st[:, :, 0].min()) / 100.0            # 100 MHz -> us
names = ["start", "loaded", "rows out", "arrived", "released", "stored", "departed", "end"]
print(f"n={n} h={h} c={c}: {bpi} blocks per sample; us from the first start (min / median / max over a sample's blocks)")
for s in list(range(min(n, 10))) + ([n - 1] if n > 10 else []):
    print(f"sample {s:2d}: " + "  ".join(f"{nm} {np.min(st[s, :, i]):6.1f}/{np.median(st[s, :, i]):6.1f}/{np.max(st[s, :, i]):6.1f}" for i, nm in enumerate(names)))
d = np.diff(st, axis=2)
print("median phase lengths (us): " + "  ".join(f"{names[i]}->{names[i + 1]} {np.median(d[:, :, i]):.1f}" for i in range(7)))
print(f"whole launch {st[:, :, 7].max():.1f} us")
