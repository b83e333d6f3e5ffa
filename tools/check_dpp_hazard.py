#!/usr/bin/env python3
"""Build-time check of the hand-written DPP instructions (rwm_row.hip).

gfx950 rule: a VGPR written by a VALU instruction must not be read through DPP (the src0 operand of a *_dpp
instruction) for two wait states.  The compiler keeps that rule for its own DPP instructions, but it does not see into
`asm` blocks: the v_fmac_f64_dpp blocks of rwm_row.hip rely on an `s_nop 1` placed where their operand is created
(dpp_ready, the RowVec constructors).  If a later compiler, flag or code change puts a register copy (a live-range split
v_mov, a v_accvgpr_read) between that fence and the asm block, the DPP read is unprotected and the kernel silently
reads a stale value.  This script disassembles the gfx950 code object inside a hipcc object file and fails when any
DPP instruction reads a VGPR that a VALU instruction wrote within the two preceding wait states (an `s_nop N` counts
N + 1; a taken branch costs more than two, so the scan follows the fall-through order of the disassembly).

Usage: tools/check_dpp_hazard.py mcmc_amd/csrc/rwm_row.o [more .o ...]     exit code 1 on a violation.
"""
from __future__ import annotations

import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_SGPR_WRITERS = ("v_readlane", "v_readfirstlane", "v_cmp", "v_cmpx")


def _regs(operand):
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def disassemble(obj_path):
    """-> list of (mnemonic, [operands]) of the gfx950 code object bundled in a hipcc .o / .so"""
    tmp = tempfile.mkdtemp(prefix="dppchk_")
    try:
        local = os.path.join(tmp, os.path.basename(obj_path))
        shutil.copy(obj_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True)
        cos = [f for f in glob.glob(local + ".*") if "amdgcn" in f]
        if not cos:
            return []
        text = subprocess.run([OBJDUMP, "-d", cos[0]], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    insts = []
    for line in text.splitlines():
        if not line.startswith("\t"):
            insts.append(("<label>", []))  # symbol / blank line: a new function starts with a clean slate
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        insts.append((parts[0], ops))
    return insts


def violations(insts):
    """[(index, mnemonic, operands, offending registers)]"""
    bad = []
    recent = []  # VGPRs written by VALU in each of the last wait states, newest last
    for i, (mn, ops) in enumerate(insts):
        if mn == "<label>":
            recent = []
            continue
        if mn.endswith("_dpp") or " row_" in " ".join(ops) or "quad_perm" in " ".join(ops):
            # dst, src0 (read through DPP), ...; the modifiers ride on the last operand
            src0 = _regs(ops[1].split(" ")[0]) if len(ops) > 1 else set()
            hot = set().union(*recent[-2:]) if recent else set()
            if src0 & hot:
                bad.append((i, mn, ops, sorted(src0 & hot)))
        if mn == "s_nop":
            n = int(ops[0], 0) + 1 if ops else 1
            recent.extend([set()] * n)
        elif mn.startswith("v_") and not mn.startswith(_SGPR_WRITERS) and ops:
            recent.append(_regs(ops[0].split(" ")[0]))
        else:
            recent.append(set())
        recent = recent[-4:]
    return bad


def check(obj_path):
    insts = disassemble(obj_path)
    n_dpp = sum(1 for mn, _ in insts if mn.endswith("_dpp"))
    return n_dpp, violations(insts)


def main(argv):
    rc = 0
    for path in argv[1:]:
        n_dpp, bad = check(path)
        print(f"{path}: {n_dpp} DPP instructions, {len(bad)} hazard violations")
        for i, mn, ops, regs in bad[:20]:
            print(f"   inst #{i}: {mn} {', '.join(ops)}   <- v{regs} written by VALU within 2 wait states")
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
