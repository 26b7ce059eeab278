#!/usr/bin/env python3
"""Scan the gfx950 code of libshmgan_hip.so for the VMEM store-data hazard:

    a store of more than 64 bits (buffer / global / flat / scratch _store_dwordx3 / x4)
    followed within two wait states by a VALU instruction that WRITES one of the store's data VGPRs

The store reads its data registers after it has issued; gfx9 wants `s_nop`s in between (hipcc's hazard recogniser: two wait states from gfx940
on).  hipcc inserts them -- EXCEPT after a buffer store whose soffset is an SGPR, which its table (from the Southern-Islands documents) exempts.
The MI355X does not: round 4's conv3x3s2_rgb_fwd_kernel had `buffer_store_dwordx4 v[16:19], v104, s[36:39], s66 offen` directly followed by a
v_max_f32 into v17 (the next tile's LeakyReLU), and ~4e-4 of that register's values were stored as the next tile's -- lanes 12-15 of a row only,
only at sizes that fill the chip, differently from run to run.  With the tile offset in the instruction's immediate field the compiler spaces
the write out itself.  The scan applies the rule to every wide store, whatever its soffset.
The scan also applies the neighbouring rule inline asm escapes the same way: no VALU read of an MFMA's VGPR result within passes + 2 wait states of
it (per MFMA shape; hipcc pads its own instructions -- since round 5 no inline asm reads an accumulator, the conversions are compiler-visible).
(Linear scan of the disassembly: a store at the very end of a loop body against a write at its top is not seen.)

    python tools/check_isa_hazards.py [path/to/libshmgan_hip.so]      exit 0 = clean, 1 = findings (printed with kernel and lines)

tests/test_abi.py runs it on the built library (no GPU needed)."""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
WAIT_STATES = 2


def code_objects(lib: Path):
    """the gfx950 ELF images inside the library's clang offload bundles"""
    blob = lib.read_bytes()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = q


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(tok):
    m = REG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


# wait states between an MFMA and a VALU read of its VGPR result: passes + 2 (a pass = 4 cycles of the matrix pipe), per instruction
MFMA_PASSES = {"v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_16x16x32_bf16": 4,
               "v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_16x16x32_f16": 4}
MFMA_DEFAULT_PASSES = 4


def mfma_wait_states(ins):
    return MFMA_PASSES.get(ins.replace("_e64", ""), MFMA_DEFAULT_PASSES) + 2


def nop_states(ins, ops):
    if ins == "s_nop":
        return int(ops[0], 0) + 1
    if ins.startswith("v_mfma") or ins.startswith("v_smfmac"):
        # an MFMA behind another one issues when the matrix pipe takes it: it holds the wave for about its passes
        return MFMA_PASSES.get(ins.replace("_e64", ""), MFMA_DEFAULT_PASSES)
    return 1


def all_vregs(ops):
    out = set()
    for o in ops:
        for m in REG.finditer(o):
            if m.group(1) is not None:
                out.add(int(m.group(1)))
            else:
                out |= set(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def scan(text):
    findings, kernel = [], "?"
    pending = []                  # [(data regs, wait states left, store line)]
    mfma = []                     # [(result VGPRs, wait states left, mfma line)]: results in arch VGPRs a VALU must not read yet
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            kernel, pending, mfma = m.group(1), [], []
            continue
        body = line.split("//")[0].strip()
        if not body or body.startswith("Disassembly") or body.endswith(":"):
            continue
        parts = body.split(None, 1)
        ins = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        # VALU write of a pending store's data?
        if ins.startswith("v_") and not ins.startswith("v_cmp") and ops:
            dst = vregs(ops[0])
            if ins.startswith("v_mfma") or ins.startswith("v_smfmac") or ins.startswith("v_accvgpr_write"):
                dst = set()       # AGPR destinations
            for regs, left, where in pending:
                if dst & regs:
                    findings.append((kernel, where, body))
        # second rule (same blind spot: hipcc counts the wait states for its own instructions, not for inline asm): a VALU instruction -- not another
        # MFMA, which the matrix pipe orders itself -- reading the VGPR result of an MFMA within MFMA_WAIT_STATES of it
        if ins.startswith("v_") and not (ins.startswith("v_mfma") or ins.startswith("v_smfmac")) and len(ops) > 1:
            src = all_vregs(ops[1:])
            for regs, left, where in mfma:
                if src & regs:
                    findings.append((kernel, where, body))
        # a VALU write to a register an MFMA result sits in supersedes that result for later readers; code behind an unconditional
        # branch is not reached by falling through
        if ins.startswith("v_") and not ins.startswith("v_cmp") and ops and not (ins.startswith("v_mfma") or ins.startswith("v_smfmac")):
            w = vregs(ops[0])
            mfma = [(r - w, left, wh) for r, left, wh in mfma if r - w]
        if ins in ("s_branch", "s_endpgm", "s_setpc_b64"):
            pending, mfma = [], []
        states = nop_states(ins, ops)
        pending = [(r, left - states, w) for r, left, w in pending if left - states > 0]
        mfma = [(r, left - states, w) for r, left, w in mfma if left - states > 0]
        if (ins.startswith("v_mfma") or ins.startswith("v_smfmac")) and ops:
            dst = vregs(ops[0])       # empty for AGPR destinations
            if dst:
                mfma.append((dst, mfma_wait_states(ins), body))
        wide = ins.endswith("dwordx3") or ins.endswith("dwordx4")
        if wide and "store" in ins:
            # buffer_store: vdata, vaddr, srsrc, soffset; global / flat / scratch: vaddr, vdata, saddr
            data = vregs(ops[0] if ins.startswith("buffer_") else ops[1])
            if data:
                pending.append((data, WAIT_STATES, body))
    return findings


def main(argv):
    lib = Path(argv[1]) if len(argv) > 1 else Path(__file__).resolve().parents[1] / "shmgan_amd" / "libshmgan_hip.so"
    objs = code_objects(lib)
    if not objs:
        print(f"{lib}: no gfx950 code object found")
        return 2
    bad = []
    nk = 0
    with tempfile.TemporaryDirectory() as td:
        for i, img in enumerate(objs):
            p = Path(td) / f"co{i}.elf"
            p.write_bytes(img)
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(p)], check=True, capture_output=True, text=True).stdout
            nk += len(re.findall(r"^[0-9a-f]+ <.+>:$", text, re.M))
            bad += scan(text)
    for kernel, store, writer in bad:
        print(f"HAZARD {kernel}\n    {store}\n    {writer}")
    print(f"{lib.name}: {len(objs)} code objects, {nk} symbols, {len(bad)} store-data / MFMA-result hazards")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
