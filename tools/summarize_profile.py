#!/usr/bin/env python3
"""Condense rocprofv3 csv output (tools/profile_bench.sh) into a small text/JSON summary."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    for key in ("k_wsplit_accum", "k_winverse", "k_w1_accum", "k_w1_bp", "k_wf_sum", "k_wf_fold", "k_wf_lags", "k_relayout",
                "k_synth", "k_unlayout", "k_bp_transpose", "k_row_sums", "k_direct", "k_sum_partials", "k_helfand"):
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
    # HBM traffic of the dominant kernel per launch, for bench.py's roofline.traffic: FETCH_SIZE is
    # in KB and reports half of the bytes of a wide streaming read on gfx950
    # (MI355X_MICROARCH.md, HBM): 2 * FETCH_SIZE + WRITE_SIZE.  Keyed by the library's hash.
    if len(sys.argv) > 3:
        key, kern = sys.argv[2], sys.argv[3]
        pmc = res.get("pmc", {}).get(kern, {})
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            import hashlib

            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            so = os.path.join(root, "transport_analysis_amd", "libta_hip.so")
            sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
            f = os.path.join(out, "hbm_traffic.json")
            rec = {"so_sha16": sha, "entries": {}}
            rec["entries"][key] = {
                "hbm_bytes_per_launch": 2 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024,
                "FETCH_SIZE_KB_raw": pmc["FETCH_SIZE"], "WRITE_SIZE_KB_raw": pmc["WRITE_SIZE"],
                "TCC_EA0_RDREQ_sum": pmc.get("TCC_EA0_RDREQ_sum"), "kernel": kern,
                "note": "2*FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate --pmc passes; L2 fabric-side "
                        "counters (Infinity-Cache hits are counted)"}
            json.dump(rec, open(f, "w"), indent=1)
            print("wrote", f, rec["entries"][key]["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
