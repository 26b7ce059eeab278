#!/usr/bin/env python3
"""CPU-side sanitizer build of the host half of libshmgan_hip.so ("race detection / sanitizers").

Every csrc/*.hip is compiled with `-fsanitize=address,undefined -fno-gpu-sanitize`: the HOST code (argument validation, dispatch, the launchers'
arithmetic, the tuning table, error strings, thread-local one-shot state) is instrumented, the device code is built as usual and never runs --
GPU AddressSanitizer is not available on this pool and is not attempted.  The instrumented library goes to $SHM_ASAN_DIR (default
/tmp/shm_asan), never into the tree.  Then the GPU-less tests that drive the C ABI run against it under the ASan runtime:

    tests/test_abi.py           header <-> exports <-> ctypes table, every entry point's argument validation and error path, the
                                thread-local one-shot state, shm_set_tuning / shm_get_tuning bounds
    tests/test_norm_queries.py  the launchers' shape arithmetic (workspace sizes, *_supported queries) over a grid of shapes

Usage:  python tools/sanitize_host.py [--jobs N] [--log FILE]     (exit code = pytest's; any ASan / UBSan report fails the run)
"""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--log", default=None)
    ap.add_argument("--tests", nargs="*", default=["tests/test_abi.py", "tests/test_norm_queries.py"])
    a = ap.parse_args()
    from shmgan_amd import _lib
    out = Path(os.environ.get("SHM_ASAN_DIR", "/tmp/shm_asan"))
    out.mkdir(parents=True, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    san = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-sanitize-recover=undefined", "-shared-libsan", "-fno-omit-frame-pointer", "-g"]
    cflags = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-O3")] + ["-O1"] + san
    shared = max(p.stat().st_mtime for p in [_lib.CSRC / h for h in ("common.h", "ablate.h", "tapgemm.h", "wgrad.h", "x3split.h")] + [_lib.HEADER])

    def cc(name):
        src, obj = _lib.CSRC / name, out / (Path(name).stem + ".o")
        if obj.exists() and obj.stat().st_mtime >= max(src.stat().st_mtime, shared):
            return obj
        cmd = [hipcc, *cflags, *_lib.EXTRA_FLAGS.get(name, []), "-c", str(src), "-o", str(obj)]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj
    with ThreadPoolExecutor(max_workers=a.jobs) as ex:
        objs = list(ex.map(cc, _lib.SOURCES))
    lib = out / "libshmgan_hip_asan.so"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *san, *map(str, objs), "-o", str(lib)], check=True)
    rt = subprocess.run([hipcc, "-print-file-name=libclang_rt.asan-x86_64.so"], check=True, capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt):
        import glob
        rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))[-1]
    env = dict(os.environ, SHM_LIB_PATH=str(lib), LD_PRELOAD=rt,
               # leaks: CPython and torch keep arenas alive at exit by design; everything else stays on and is fatal
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    cmd = [sys.executable, "-m", "pytest", *a.tests, "-q", "-m", "not gpu", "-p", "no:cacheprovider"]
    print("SHM_LIB_PATH=%s LD_PRELOAD=%s %s" % (lib, rt, " ".join(cmd)), flush=True)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    text = r.stdout + r.stderr
    bad = [ln for ln in text.splitlines() if "AddressSanitizer" in ln or "runtime error:" in ln]
    print(text[-6000:])
    verdict = f"sanitize_host: pytest exit {r.returncode}, {len(bad)} sanitizer report line(s)"
    print(verdict)
    if a.log:
        Path(a.log).write_text("\n".join(["$ python tools/sanitize_host.py", "flags: " + " ".join(cflags), "library: " + str(lib), "runtime: " + rt,
                                          "tests: " + " ".join(a.tests), "", text[-12000:], verdict, ""]))
    return 1 if (r.returncode != 0 or bad) else 0


if __name__ == "__main__":
    sys.exit(main())
