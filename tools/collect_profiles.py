#!/usr/bin/env python3
"""Copies what tools/profile_all.sh TAG left under gpurun_out/ into profiles/ (tracked):
profiles/<TAG>_<case>_summary.txt (the rocprofv3 kernel stats and PMC means, headed by the
library hash and the hipEvent medians of the SAME profiled run), <TAG>_<case>_kernel_stats.csv,
and profiles/hbm_traffic.json (bench.py's `roofline.traffic`, valid for that library hash only).

    python tools/collect_profiles.py r04
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    go = os.path.join(ROOT, "gpurun_out")
    traffic = json.load(open(os.path.join(go, f"hbm_traffic_{tag}.json")))
    sha = traffic["so_sha16"]
    for d in sorted(glob.glob(os.path.join(go, f"prof_{tag}_*"))):
        if not os.path.isdir(d):
            continue
        case = os.path.basename(d)[len(f"prof_{tag}_"):]
        summ = os.path.join(d, "summary.txt")
        if not os.path.exists(summ):
            continue
        head = [f"# rocprofv3 summary of: python bench.py (tools/profile_all.sh {tag}, case {case}), library sha16 {sha}"]
        try:  # the bench line of the --stats run: hipEvents of the same run
            line = [l for l in open(os.path.join(d, "bench_stats.log")) if l.startswith("{")][-1]
            b = json.loads(line)
            head.append("# hipEvents of the SAME profiled run: ms_per_step %.4f, whole call median %.4f ms, "
                        "dominant kernel median %.4f ms" % (b["ms_per_step"], b["device_ms"]["whole_call_median"],
                                                            b["device_ms"]["dominant_kernel_median"]))
        except Exception as e:
            head.append(f"# (no bench line found in bench_stats.log: {e})")
        with open(os.path.join(ROOT, "profiles", f"{tag}_{case}_summary.txt"), "w") as f:
            f.write("\n".join(head) + "\n" + open(summ).read())
        for st in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(st, os.path.join(ROOT, "profiles", f"{tag}_{case}_kernel_stats.csv"))
        print("profiles/%s_%s_summary.txt" % (tag, case))
    shutil.copy(os.path.join(go, f"hbm_traffic_{tag}.json"), os.path.join(ROOT, "profiles", "hbm_traffic.json"))
    print("profiles/hbm_traffic.json for library", sha)


if __name__ == "__main__":
    main()
