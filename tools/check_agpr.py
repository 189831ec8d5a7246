#!/usr/bin/env python3
"""Build-time check of the manual AGPR file (fft_kernels.hpp).  k_fft_accum keeps the
gathered pair, the landing zone and (smaller plans) an accumulator set in hand-assigned
AGPRs [agpr_base<P>(), 256), touched only from inline asm.  The compiler cannot be told
that those registers are live; under pressure it parks values of its own in AGPRs it
believes free.  The kernel announces its range with an assembler comment
("; TA_AGPR_MANUAL_RANGE lo hi"), and every instruction OUTSIDE an inline-asm block (LLVM
brackets those with ;;#ASMSTART / ;;#ASMEND) whose destination is an AGPR inside that
range is the compiler writing into the manual range: the build fails.
Usage: check_agpr.py file.s [...]"""
import re
import sys

bad = kernels = 0
dst_re = re.compile(r"^\s*([a-z_0-9]+)\s+a(?:\[(\d+)(?::(\d+))?\]|(\d+))\b")
for path in sys.argv[1:]:
    name, base, top, in_asm = None, None, 256, False
    for line in open(path):
        m = re.match(r"^(_ZN2ta11k_fft_accum\S+):", line)
        if m:
            name, base, top, in_asm = m.group(1), None, 256, False
            kernels += 1
            continue
        if name is None:
            continue
        if "s_endpgm" in line:
            if base is None:
                print(f"{path}: {name}: no TA_AGPR_MANUAL_RANGE marker")
                bad += 1
            name = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        mk = re.search(r"TA_AGPR_MANUAL_RANGE (\w+) (\w+)", line)
        if mk:
            base, top = int(mk.group(1), 0), int(mk.group(2), 0)  # large immediates print as hex
            continue
        if in_asm or base is None:
            continue
        d = dst_re.match(line)
        if d and not d.group(1).startswith(("global_store", "ds_write", "scratch_store", "buffer_store")):
            hi = int(d.group(3) or d.group(2) or d.group(4))
            lo_reg = int(d.group(2) or d.group(4))
            if hi >= base and lo_reg < top:
                print(f"{path}: {name}: compiler-owned '{line.strip()}' writes into the manual range a{base}..a{top - 1}")
                bad += 1
print("agpr check:", "FAILED" if bad else "ok", f"({kernels} kernels, {bad} offending instructions)")
sys.exit(1 if bad else 0)
