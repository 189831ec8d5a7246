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
                "k_synth", "k_unlayout", "k_bp_transpose", "k_row_sums", "k_direct", "k_sum_partials",
                "k_helfand_product32", "k_helfand_product", "k_helfand_combine", "k_band32_tp", "k_band32_lags", "k_band32_bp", "k_widen_f32", "k_helfand",
                "k_band_bp_vacf", "k_band_bp_helf", "k_bandbp_gather", "k_band_lags", "k_band_gather", "k_short", "k_mid"):
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
        import hashlib

        key, kern = sys.argv[2], sys.argv[3]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        so = os.path.join(root, "transport_analysis_amd", "libta_hip.so")
        sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
        f = os.path.join(out, "hbm_traffic.json")
        rec = {"so_sha16": sha, "entries": {}}
        # entries of earlier runs with the same library are kept (TA_TRAFFIC_MERGE = their file)
        prev = os.environ.get("TA_TRAFFIC_MERGE", "")
        if prev and os.path.exists(prev):
            old = json.load(open(prev))
            if old.get("so_sha16") == sha:
                rec["entries"].update(old.get("entries", {}))
        note = ("2*FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate --pmc passes; L2 fabric-side counters "
                "(Infinity-Cache hits are counted)")
        if "+" not in kern:
            pmc = res.get("pmc", {}).get(kern, {})
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                rec["entries"][key] = {
                    "hbm_bytes_per_launch": 2 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024,
                    "FETCH_SIZE_KB_raw": pmc["FETCH_SIZE"], "WRITE_SIZE_KB_raw": pmc["WRITE_SIZE"],
                    "TCC_EA0_RDREQ_sum": pmc.get("TCC_EA0_RDREQ_sum"), "kernel": kern, "note": note}
        else:
            # a path of several kernels per step: every dispatch of every named kernel, summed, per
            # step of the profiled command (TA_PROFILE_STEPS = its steps + warm-ups)
            n_steps = int(os.environ.get("TA_PROFILE_STEPS", "4"))
            tot, parts = 0.0, {}
            for sub, cname, mul in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
                for fcsv in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
                    for r in csv.DictReader(open(fcsv)):
                        k = short(r["Kernel_Name"])
                        if k in kern.split("+") and r["Counter_Name"] == cname:
                            b = mul * float(r["Counter_Value"]) * 1024 / n_steps
                            tot += b
                            parts[k] = parts.get(k, 0.0) + b
            if tot > 0:
                rec["entries"][key] = {"hbm_bytes_per_launch": tot, "per_kernel_bytes_per_step": parts,
                                       "kernel": kern, "steps_profiled": n_steps, "note": note + "; per step"}
        if rec["entries"]:
            json.dump(rec, open(f, "w"), indent=1)
            print("wrote", f, {k: v["hbm_bytes_per_launch"] for k, v in rec["entries"].items()})


if __name__ == "__main__":
    main()
