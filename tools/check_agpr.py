#!/usr/bin/env python3
"""Build-time check: in every VEC accumulate kernel the only v_accvgpr_write the code may
contain are the manual accumulator slots (a160..a255).  Any write to a0..a159 would be
the compiler parking a value of its own on top of the gathered pair.  Usage:
check_agpr.py file.s [...]"""
import re
import sys

bad = 0
for path in sys.argv[1:]:
    name = None
    for line in open(path):
        m = re.match(r"^(_ZN2ta11k_fft_accum\S+):", line)
        if m:
            name = m.group(1)
        if line.startswith("\ts_endpgm"):
            name = None
        if name and "Lb1ELb" in name:  # VEC = true instantiations
            w = re.match(r"\s*v_accvgpr_write_b32 a(\d+),", line)
            if w and int(w.group(1)) < 160:
                print(f"{path}: {name}: compiler-owned write to a{w.group(1)}")
                bad += 1
print("agpr check:", "FAILED" if bad else "ok", f"({bad} offending writes)")
sys.exit(1 if bad else 0)
