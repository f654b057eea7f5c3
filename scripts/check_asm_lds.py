#!/usr/bin/env python3
"""Build-time guard for the LDS reads the anneal kernels issue from inline asm (csrc/sparse_kernels.hip,
sparse_pair_kernels.hip, sparse_split_kernels.hip, potts_fast_kernels.hip).

hipcc's s_waitcnt pass does not count instructions inside inline asm, so those kernels wait for their own reads with an
asm `s_waitcnt lgkmcnt(N)` and pass every destination register THROUGH that statement ("+v"): "used only after the wait"
is a data dependence of the C++ source.  This script checks the EMITTED code for the same property: it compiles each
file to gfx950 assembly and walks every kernel linearly; a VGPR written by an asm `ds_read_*` is PENDING until a
`s_waitcnt lgkmcnt(N)` (asm or compiler-made; LDS reads return in order, so after lgkmcnt(N) only the N youngest can
still be in flight) retires it, and no instruction may name a pending register as an operand before that.

    python scripts/check_asm_lds.py            (exit code 0 = every kernel clean; run by __graft_entry__.build())
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scrna_seq_qannealing_clustering_amd", "csrc")
FILES = ["sparse_kernels.hip", "sparse_pair_kernels.hip", "sparse_split_kernels.hip", "potts_fast_kernels.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "--offload-arch=gfx950",
         "--offload-device-only", "-S"]
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_asm(path):
    """Returns (kernels seen, asm LDS reads seen, list of violations)."""
    kernels = reads = 0
    bad = []
    pending = []            # [(vgpr, line number of the read)] in issue order
    in_asm = False
    name = None
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.startswith("_Z") and raw.rstrip().endswith(":") or re.match(r"^_Z\w+:", raw):
            name, pending, kernels = raw.split(":")[0], [], kernels + 1
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        op, _, rest = line.partition(" ")
        wait = re.search(r"lgkmcnt\((\d+)\)", line) if op == "s_waitcnt" else None
        if wait:
            keep = int(wait.group(1))
            pending = pending[len(pending) - keep:] if keep else []
            continue
        if in_asm and op.startswith("ds_read"):
            dst, _, src = rest.partition(",")
            used = regs_of(src)
            hit = [p for p in pending if p[0] in used]
            if hit:
                bad.append((name, ln, raw.strip(), hit))
            for r in sorted(regs_of(dst)):
                pending.append((r, ln))
            reads += 1
            continue
        used = regs_of(rest)
        hit = [p for p in pending if p[0] in used]
        if hit and not op.startswith("s_"):
            bad.append((name, ln, raw.strip(), hit))
    return kernels, reads, bad


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    failed = False
    with tempfile.TemporaryDirectory() as tmp:
        for f in FILES:
            out = os.path.join(tmp, f + ".s")
            subprocess.run([hipcc] + FLAGS + ["-o", out, os.path.join(CSRC, f)], check=True, stderr=subprocess.DEVNULL)
            kernels, reads, bad = check_asm(out)
            print("%-28s %2d kernels, %4d asm LDS reads, %d violations" % (f, kernels, reads, len(bad)))
            for name, ln, text, hit in bad[:10]:
                failed = True
                print("   %s line %d: `%s` names v%d, read from LDS at line %d and not yet waited for"
                      % (name, ln, text, hit[0][0], hit[0][1]))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
