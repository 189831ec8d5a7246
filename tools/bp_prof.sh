#!/bin/bash
# Run on the GPU box: per-kernel durations of the by-particle FFT evaluation (tools/bp_ab.py).
#   tools/bp_prof.sh [n_frames n_atoms dim spec_atoms]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
OUT=$R/gpurun_out/bp_prof
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/bp_ab.py ${@:-10000 100000 3} > $OUT/run.log 2>&1
find $OUT -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "%9.4f ms" % (float(r["AverageNs"]) / 1e6), r["Percentage"])
PY
grep -v amdgpu.ids $OUT/run.log
