#!/usr/bin/env python3
"""Disassembly of one kernel of libshmgan_hip.so and its instruction mix.

    python tools/kernel_isa.py <substring of the mangled or demangled symbol> [--lib path] [--dump out.s] [--loop]

Prints, per matching kernel, the count of instructions by issue class (MFMA, other VALU, LDS, vector memory, scalar, waits) -- the
numbers the vector-issue budget of an MFMA-paced loop is made of (MI355X guide: an MFMA holds the SIMD's vector issue for 8 cycles,
every other vector instruction for 4) -- and with --dump writes the disassembly.  --loop restricts the count to the largest
backward-branch body (the patch loop of the persistent kernels)."""
import argparse
import re
import subprocess
import sys
import tempfile
from collections import Counter
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from check_isa_hazards import OBJDUMP, code_objects  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent


def classify(ins):
    if ins.startswith("v_mfma") or ins.startswith("v_smfma"):
        return "mfma"
    if ins.startswith("ds_"):
        return "lds"
    if ins.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if ins.startswith("v_"):
        return "valu"
    if ins.startswith("s_waitcnt") or ins.startswith("s_nop") or ins.startswith("s_barrier"):
        return "wait"
    if ins.startswith("s_"):
        return "salu"
    return "other"


def kernels(lib):
    for blob in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "-C", f.name], capture_output=True, text=True, check=True).stdout
        name, body = None, []
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                if name is not None:
                    yield name, body
                name, body = m.group(1), []
            elif name is not None and line.strip():
                body.append(line)
        if name is not None:
            yield name, body


def parse(body):
    """[(address or None, mnemonic, operand text)] -- objdump prints '\tins ops // ADDR: ' comments"""
    out = []
    for line in body:
        s = line.strip()
        m = re.match(r"^(\S+)\s*(.*?)\s*(?://\s*([0-9A-Fa-f]+):(.*))?$", s)
        if not m:
            continue
        out.append((int(m.group(3), 16) if m.group(3) else None, m.group(1), m.group(2) + " " + (m.group(4) or "")))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pattern")
    ap.add_argument("--lib", default=str(ROOT / "shmgan_amd" / "libshmgan_hip.so"))
    ap.add_argument("--dump", default="")
    ap.add_argument("--loop", action="store_true")
    args = ap.parse_args()
    found = 0
    for name, body in kernels(Path(args.lib)):
        if args.pattern not in name:
            continue
        found += 1
        ins = parse(body)
        lo, hi = 0, len(ins)
        if args.loop:
            addr = {a: i for i, (a, _, _) in enumerate(ins) if a is not None}
            best = (0, 0, 0)
            for i, (a, m, ops) in enumerate(ins):
                if m.startswith("s_cbranch") or m == "s_branch":
                    t = re.search(r"\+0x([0-9a-fA-F]+)>\s*$", ops)
                    if t:
                        base = ins[0][0] if ins[0][0] is not None else 0
                        j = addr.get(base + int(t.group(1), 16))
                        if j is not None and j < i and i - j > best[0]:
                            best = (i - j, j, i + 1)
            if best[0]:
                lo, hi = best[1], best[2]
        c = Counter(classify(m) for _, m, _ in ins[lo:hi])
        vec = c["valu"] + c["lds"] + c["vmem"]
        print(f"{name}\n  instructions {hi - lo}{' (largest loop)' if args.loop else ''}: " + ", ".join(f"{k} {c[k]}" for k in ("mfma", "valu", "lds", "vmem", "salu", "wait", "other"))
              + f"\n  vector issue cycles: {8 * c['mfma']} (MFMA) + {4 * vec} (other vector) = {8 * c['mfma'] + 4 * vec}; non-MFMA vector instructions {vec}")
        top = Counter(m for _, m, _ in ins[lo:hi] if classify(m) in ("valu", "lds", "vmem")).most_common(14)
        print("  " + ", ".join(f"{m} {n}" for m, n in top))
        if args.dump:
            Path(args.dump).write_text("\n".join(body) + "\n")
    return 0 if found else 1


if __name__ == "__main__":
    sys.exit(main())
