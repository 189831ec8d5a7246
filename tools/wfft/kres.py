"""kernel-resource-usage remarks of hipcc -> one line per wfft kernel (name, VGPRs, scratch, occupancy)"""
import re
import sys

name, rows, cur = None, [], None
for l in sys.stdin:
    if "error" in l:
        print(l.rstrip())
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        name = m.group(1)
        cur = {"n": name}
        rows.append(cur)
    for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, l)
        if m and cur is not None:
            cur[key] = m.group(1)
for r in rows:
    if not re.search(r"k_w|k_band", r["n"]):
        continue
    short = re.sub(r"_ZN2ta\d+", "", r["n"])
    short = re.sub(r"INS_5WPlanILi(\d+)EEE", r"<R0=\1>", short)[:48]
    print("%-50s VGPR %s scratch %s occ %s" % (short, r.get("vgpr"), r.get("scratch"), r.get("occ")))
