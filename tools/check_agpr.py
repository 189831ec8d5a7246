#!/usr/bin/env python3
"""Build-time check of the manual AGPR file (fft_kernels.hpp).  The manual slots sit at the
top of the AGPR file: the gathered pair (only ever written by global loads in VEC kernels:
global_load_dwordx4 a[N:N+3]) followed by the accumulator slots (the quads stored with
global_store_dwordx4 ... a[N:N+3]).  A v_accvgpr_write into [lowest loaded quad, lowest
stored quad) would be the compiler parking a value of its own on top of the gathered pair.
Usage: check_agpr.py file.s [...]"""
import re
import sys

bad = 0
kernels = 0
for path in sys.argv[1:]:
    name, writes, stores, loads = None, [], [], []

    def finish():
        global bad, kernels
        if name and "Lb1ELb" in name:  # VEC = true instantiations
            kernels += 1
            if not stores:
                print(f"{path}: {name}: no accumulator stores found")
                bad += 1
                return
            base = min(stores)
            lo = min(loads) if loads else base
            for w in writes:
                if lo <= w < base:
                    print(f"{path}: {name}: compiler-owned write to a{w} (gathered pair a{lo}..a{base - 1})")
                    bad += 1

    for line in open(path):
        m = re.match(r"^(_ZN2ta11k_fft_accum\S+):", line)
        if m:
            finish()
            name, writes, stores, loads = m.group(1), [], [], []
        w = re.match(r"\s*v_accvgpr_write_b32 a(\d+),", line)
        if w and name:
            writes.append(int(w.group(1)))
        st = re.match(r"\s*global_store_dwordx4 v\[\d+:\d+\], a\[(\d+):", line)
        if st and name:
            stores.append(int(st.group(1)))
    finish()
print("agpr check:", "FAILED" if bad else "ok", f"({kernels} kernels, {bad} offending writes)")
sys.exit(1 if bad else 0)
