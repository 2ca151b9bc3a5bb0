#!/usr/bin/env python3
"""Reads the gfx950 assembly of csrc/xee_sweepz.hip and checks what the hardware does not: a register that an LDS read
(ds_read2st64_b32, issued by k_sweep_z's inline assembly a tick ahead of its use) is still filling must not be touched by
any instruction before the next `s_waitcnt lgkmcnt(...)`.  The compiler does not know these registers are pending and may
copy them (a move in front of the waiting statement's tied operand); this scan is how such a build is refused.
    python tools/isa_lds_hazards.py [file.s]     (without a file: compiles the kernel with the Makefile's flags first)
Prints the hazards found; exit status 1 if any."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def compile_asm():
    mk = (ROOT / "xlab-ee-fortran_amd" / "Makefile").read_text().replace("\\\n", " ")
    flags = re.search(r"^HIPFLAGS\s*:=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    out = Path(tempfile.mkdtemp()) / "xee_sweepz.s"
    cmd = ["/opt/rocm/bin/hipcc", *flags, "-fno-slp-vectorize", "--cuda-device-only", "-S", "-I", str(ROOT / "include"),
           "-I", str(ROOT / "xlab-ee-fortran_amd" / "csrc"), str(ROOT / "xlab-ee-fortran_amd" / "csrc" / "xee_sweepz.hip"), "-o", str(out)]
    subprocess.check_call(cmd)
    return out


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def scan(path):
    pending, hazards, reads = {}, [], 0
    for n, line in enumerate(Path(path).read_text().split("\n"), 1):
        t = line.strip()
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if op.endswith(":"):
            continue
        args = [a.strip().split(" ")[0] for a in t[len(op):].split(",")]
        if op == "s_waitcnt" and "lgkmcnt" in t:
            pending.clear()
            continue
        if op == "ds_read2st64_b32":
            reads += 1
            for r in regs(args[0]):
                pending[r] = n
            args = args[1:]                 # its address operand is an ordinary use
        for a in args:
            for r in regs(a):
                if r in pending:
                    hazards.append((n, t, pending[r]))
    return reads, hazards


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_asm()
    reads, hazards = scan(path)
    print(f"{reads} ds_read2st64_b32, {len(hazards)} uses of a pending register")
    for n, t, src in hazards[:20]:
        print(f"  line {n}: {t}   (pending since line {src})")
    sys.exit(1 if hazards or reads == 0 else 0)
