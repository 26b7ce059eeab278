"""Timing-only ablations of the tap-GEMM and weight-gradient kernels (outputs are WRONG in the ablated builds): builds
variants of the library into /tmp with -DSHM_ABL_* and runs tools/bench_conv.py against each through
SHM_LIB_PATH.  Usage: python tools/ablate_conv.py [variant ...] -- [bench_conv shapes ...]

The switches live in shmgan_amd/csrc/ablate.h (the only file that looks at the macros; the kernels use `if constexpr (abl::x)`, so
both sides of every fork are compiled in every build).  Each ablated library must export exactly the product's C ABI: asserted
below against the header's declarations and the product build's dynamic symbol table."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from shmgan_amd import _lib

VARIANTS = {"base": [], "nostore": ["-DSHM_ABL_NOSTORE"], "nomfma": ["-DSHM_ABL_NOMFMA"], "nolds": ["-DSHM_ABL_NOLDS"],
            "nolds_nodma": ["-DSHM_ABL_NOLDS", "-DSHM_ABL_NODMA"], "nodma": ["-DSHM_ABL_NODMA"],
            "fixaddr": ["-DSHM_ABL_FIXADDR"], "sameline": ["-DSHM_ABL_SAMELINE"], "noepi": ["-DSHM_ABL_NOEPI"],
            "nomfma_noepi": ["-DSHM_ABL_NOMFMA", "-DSHM_ABL_NOEPI"], "nolds_noepi": ["-DSHM_ABL_NOLDS", "-DSHM_ABL_NOEPI"],
            "nolds_nodma_noepi": ["-DSHM_ABL_NOLDS", "-DSHM_ABL_NODMA", "-DSHM_ABL_NOEPI"],
            "prio": ["-DSHM_WREG_PRIO"],
            # weight gradient (wgrad_kernel): no barrier / no global loads / no LDS stores
            "nobar": ["-DSHM_ABL_NOBAR"], "noload": ["-DSHM_ABL_NOLOAD"]}
args = sys.argv[1:]
shapes = []
bench = "bench_conv.py"
if "--variants" in args:                 # run tools/bench_variants.py (forced tap-GEMM variants) instead of bench_conv.py
    bench = "bench_variants.py"
    args.remove("--variants")
if "--" in args:
    shapes = args[args.index("--") + 1:]
    args = args[:args.index("--")]
names = args or list(VARIANTS)
for name in names:
    so = f"/tmp/libshm_abl_{name}.so"
    subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *_lib.HIPCC_FLAGS, *VARIANTS[name],
                    *[str(_lib.CSRC / s) for s in _lib.SOURCES], "-o", so], check=True)
    exported = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout.splitlines()
                if l.split()[-1].startswith("shm_")}
    product = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", str(_lib.LIB_PATH)], check=True, capture_output=True, text=True).stdout.splitlines()
               if l.split()[-1].startswith("shm_")} if _lib.LIB_PATH.exists() else set(_lib.header_functions())
    assert exported == product and set(_lib.header_functions()) <= exported, (name, sorted(exported ^ product))
    print(f"==== {name}", flush=True)
    env = dict(os.environ, SHM_LIB_PATH=so)
    subprocess.run([sys.executable, str(ROOT / "tools" / bench), *shapes], env=env, check=True)
