"""Phase stamps of tapgemm_pp_bf16_kernel (timing-only build of conv_pingpong.hip with -DSHM_ABL_STAMP, loaded through SHM_LIB_PATH): cycles per
patch and wave between the segment boundaries of the ping-pong loop, for one block in the middle of the grid.  The stamped build dumps behind the
64 bias values (this script passes a longer bias buffer).  SHM_PP_DBG = bits of timing-only switches of that build: 1 / 2 wave priority for the
X / Y segment, 4 no halo DMA after the first, 8 no epilogue, 16 no store pass (28 = the X segment beside an idle partner); --abl=NOLDS: no fragment reads.

    python tools/probes/pp_stamps.py [n,h,cin,cout]      (builds the stamped library into build_ab/ first if it is missing)"""
import os
import subprocess
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
abl = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("--abl=")]          # e.g. --abl=NOLDS: also -DSHM_ABL_NOLDS
so = ROOT / "build_ab" / ("libshm_pp_stamp" + "".join("_" + x.lower() for x in abl) + ".so")
if "SHM_LIB_PATH" not in os.environ:
    from shmgan_amd import _lib
    if not so.exists() or so.stat().st_mtime < (_lib.CSRC / "conv_pingpong.hip").stat().st_mtime:
        so.parent.mkdir(exist_ok=True)
        _lib.build()
        obj = so.parent / (so.stem + ".o")
        flags = [f for f in _lib.HIPCC_FLAGS if f != "-shared"]
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, *_lib.EXTRA_FLAGS.get("conv_pingpong.hip", []), "-DSHM_ABL_STAMP", *["-DSHM_ABL_" + x for x in abl], "-c", str(_lib.CSRC / "conv_pingpong.hip"), "-o", str(obj)], check=True)
        objs = [str(obj) if s == "conv_pingpong.hip" else str(_lib.CSRC / "_obj" / (Path(s).stem + ".o")) for s in _lib.SOURCES]
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", str(so)], check=True)
    if "--build-only" in sys.argv:
        sys.exit(0)
    os.environ["SHM_LIB_PATH"] = str(so)
    sys.exit(subprocess.run([sys.executable, *sys.argv], env=os.environ).returncode)
import torch
from shmgan_amd import ops

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n, h, cin, cout = (int(v) for v in (args[0] if args else "40,256,64,64").split(","))
dt = torch.bfloat16
x = torch.randn((n, h, h, cin), device="cuda").to(dt)
w = torch.randn((3, 3, cin, cout), device="cuda") * 0.05
wk = torch.zeros(9 * cout * cin, device="cuda", dtype=dt)
ops.transpose_taps(w, wk, 9, cin, cout, cin)
b = torch.zeros(64 + 8 * 16, device="cuda")
y = torch.empty((n, h, h, cout), device="cuda", dtype=dt)
stats = torch.empty(n * cout * 2, dtype=torch.float64, device="cuda")
scr = torch.zeros(ops.STATS_SLOTS * n * cout * 2, dtype=torch.float64, device="cuda")
ops.set_tuning("tapgemm.variant", "wreg")
sustained = "--sustained" in sys.argv          # 40 launches back to back (the chip at the clock it holds under this load), stamps of the last one
for _ in range(5):
    b.zero_()
    ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, b, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    torch.cuda.synchronize()
if sustained:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, b, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    e0.record()
    for _ in range(19):
        ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, b, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    b.zero_()
    ops.conv2d_in_fwd(x, None, 0, cin, 0, wk, b, y, cout, n, h, h, cin, cout, 3, 1, 0.2, stats, 1e-6, scratch=scr)
    e1.record()
    torch.cuda.synchronize()
    print(f"sustained: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (conv + finalize), stamps of the last launch:")
print(ops.last_kernel())
d = b[64:].view(torch.int32).cpu().numpy().reshape(8, 16)
names = ["X half 1", "mid-X bar", "X half 2", "end-X bar", "DMA issue", "epilogue", "stg bar", "stores", "stats", "halo wait", "end-Y bar"]
order = [0, 1, 2, 3, 4, 9, 5, 6, 10, 7, 8]
print("patches per group:", d[:, 11])
print("wave  " + "  ".join(f"{s:>10s}" for s in names) + "       total")
for wv in range(8):
    per = d[wv, order] / max(d[wv, 11], 1)
    print(f"{wv:4d}  " + "  ".join(f"{v:10.0f}" for v in per) + f"  {per.sum():10.0f}")
print("cycles before the loop / in the loop:", d[0, 12], d[0, 13], " 10-ns ticks:", d[0, 14], d[0, 15],
      f" -> clock {d[0, 13] / max(d[0, 15], 1) * 0.1:.3f} GHz, loop {d[0, 15] * 0.01:.1f} us, prologue {d[0, 14] * 0.01:.1f} us")
