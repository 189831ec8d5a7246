#!/usr/bin/env python3
"""Condense rocprofv3 csv output (tools/profile_bench.sh) into a small text/JSON summary."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    for key in ("k_fft_accum", "k_fft_finalize", "k_row_sums", "k_direct", "k_sum_partials",
                "k_widen_f32"):
        if key in name:
            return key
    return name[:60]


def main():
    out = sys.argv[1]
    res = {}
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
        for r in csv.DictReader(open(f)):
            print(f"{short(r['Name']):24s} calls={r['Calls']:>6s} total_ns={r['TotalDurationNs']:>14s} "
                  f"avg_ns={float(r['AverageNs']):14.1f} pct={r['Percentage']}")
            res.setdefault("stats", {})[short(r["Name"])] = {
                "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                "total_ns": int(float(r["TotalDurationNs"]))}
    for sub in ("pmc_fetch", "pmc_write", "pmc_tcc"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if not k.startswith("k_"):
                continue
            for c, v in cs.items():
                mean = sum(v) / len(v)
                print(f"{sub:10s} {k:20s} {c:22s} n={len(v):3d} mean/launch={mean:.6g}")
                res.setdefault("pmc", {}).setdefault(k, {})[c] = mean
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
