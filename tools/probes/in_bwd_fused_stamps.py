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
cb = min(c, 64)
bpi = h * h * cb // 16384
ng = n * (c // cb)
base = ops.in_bwd_fused_doubles(n, h * h, c)
scratch = torch.zeros(base + ng * bpi * 12 + 8, dtype=torch.float64, device="cuda")
for _ in range(4):
    scratch[base:].zero_()
    ops.in_bwd(g, c, None, 0, a, c, stats, red, dz, c, db, n, h, h, c, 0.2, fused=scratch)
    assert "fused" in ops.last_kernel()
torch.cuda.synchronize()
raw = scratch[base:base + ng * bpi * 12].view(torch.int64).cpu().numpy().reshape(ng, bpi, 12).astype(np.float64)
t0 = raw[:, :, 0].min()
st = (raw[:, :, :8] - t0) / 100.0            # 100 MHz -> us
names = ["start", "loaded", "rows out", "arrived", "released", "stored", "departed", "end"]
print(f"n={n} h={h} c={c}: {ng} barrier groups of {bpi} blocks; us from the first start (min / median / max over a group's blocks)")
for s in list(range(min(ng, 6))) + ([ng - 1] if ng > 6 else []):
    print(f"group {s:3d}: " + "  ".join(f"{nm} {np.min(st[s, :, i]):6.1f}/{np.median(st[s, :, i]):6.1f}/{np.max(st[s, :, i]):6.1f}" for i, nm in enumerate(names)))
d = np.diff(st, axis=2)
print("median phase lengths (us): " + "  ".join(f"{names[i]}->{names[i + 1]} {np.median(d[:, :, i]):.1f}" for i in range(7)))
# the last arriver's chain: its arrival -> row sums start -> row sums done -> means acknowledged; the others' release after that
la = raw[:, :, 8] > 0
chain = []
for gi in range(ng):
    b = np.nonzero(la[gi])[0]
    if len(b) != 1:
        continue
    r = raw[gi, b[0]]
    rel = np.median(raw[gi, ~la[gi], 4]) if bpi > 1 else r[10]
    chain.append([(r[2] - np.median(raw[gi, :, 2])) / 100, (r[8] - r[2]) / 100, (r[9] - r[8]) / 100, (r[10] - r[9]) / 100, (rel - r[10]) / 100])
chain = np.array(chain)
print("last arriver (median us): behind the median block's rows " + "  ".join(f"{nm} {v:.1f}" for nm, v in zip(
    ["", "-> arrival counted + barrier", "-> row sums", "-> means acknowledged", "-> the others released"], np.median(chain, axis=0))))
print(f"whole launch {st[:, :, 7].max():.1f} us")
