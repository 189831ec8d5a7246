#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bp_prof2
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --by-particle --frames $1 --atoms $2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --no-check > $OUT/run.log 2>&1
find $OUT -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "%9.4f ms" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
PY
