#!/usr/bin/env python3
"""Build-time check of the manual AGPR file (fft_kernels.hpp).  k_fft_accum keeps the
gathered pair, the landing zone and (smaller plans) an accumulator set in hand-assigned
AGPRs [agpr_base<P>(), 256), touched only from inline asm.  The compiler cannot be told
that those registers are live; under pressure it parks values of its own in AGPRs it
believes free.  The kernel announces its range with an assembler comment
("; TA_AGPR_MANUAL_RANGE lo hi"), and every instruction OUTSIDE an inline-asm block (LLVM
brackets those with ;;#ASMSTART / ;;#ASMEND) whose destination is an AGPR inside that
range is the compiler writing into the manual range: the build fails.  So does any such
instruction that READS an AGPR of the range (the compiler has no business there at all), and
any kernel that uses the slot accessors (v_accvgpr_* inside an asm block) without announcing
a range.  `--validated-with <hipcc version>` (the Makefile passes the toolchain the scheme was
last validated with) prints a warning when the running hipcc differs: the check itself is the
guard, the version is the reminder to re-read the generated code after a ROCm bump.
Usage: check_agpr.py [--validated-with VER --hipcc-version VER] file.s [...]"""
import re
import sys

args = sys.argv[1:]
if "--validated-with" in args:
    i = args.index("--validated-with")
    want = args[i + 1]
    del args[i:i + 2]
    have = ""
    if "--hipcc-version" in args:
        j = args.index("--hipcc-version")
        have = args[j + 1]
        del args[j:j + 2]
    if want not in have:
        print(f"agpr check: WARNING: manual AGPR scheme validated with hipcc {want}, building with '{have}': "
              "re-inspect the generated ISA of k_fft_accum (tools/check_agpr.py only sees what it knows to look for)")
bad = kernels = 0
src_re = re.compile(r"\ba(?:\[(\d+)(?::(\d+))?\]|(\d+))\b")
dst_re = re.compile(r"^\s*([a-z_0-9]+)\s+a(?:\[(\d+)(?::(\d+))?\]|(\d+))\b")
for path in args:
    name, base, top, in_asm, uses_slots = None, None, 256, False, False
    for line in open(path):
        m = re.match(r"^(_ZN2ta\d+k_\S+):", line)  # every kernel of the library's namespace
        if m:
            name, base, top, in_asm, uses_slots = m.group(1), None, 256, False, False
            continue
        if name is None:
            continue
        if "s_endpgm" in line:
            if uses_slots:
                kernels += 1
                if base is None:
                    print(f"{path}: {name}: uses the AGPR slot accessors without a TA_AGPR_MANUAL_RANGE marker")
                    bad += 1
            name = None
            continue
        if in_asm and "accvgpr" in line:
            uses_slots = True
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
                continue
        # any other mention of an AGPR of the range outside inline asm: a compiler-owned read
        ops = line.split(";")[0]
        for sm in src_re.finditer(ops):
            lo_reg = int(sm.group(1) or sm.group(3))
            hi = int(sm.group(2) or sm.group(1) or sm.group(3))
            if hi >= base and lo_reg < top and not ops.strip().startswith("."):
                print(f"{path}: {name}: compiler-owned '{line.strip()}' touches the manual range a{base}..a{top - 1}")
                bad += 1
                break
print("agpr check:", "FAILED" if bad else "ok", f"({kernels} kernels, {bad} offending instructions)")
sys.exit(1 if bad else 0)
