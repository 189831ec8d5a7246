#!/bin/bash
# second PMC set for the forward kernel of the harness: LDS queue state and latencies
# usage: pmc2.sh TAG [BINARY-SUFFIX] [harness args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; shift
BIN=$R/tools/wfft/wfft_test; if [ -n "$1" ] && [ -x "$R/tools/wfft/wfft_test_$1" ]; then BIN=$R/tools/wfft/wfft_test_$1; shift; fi
OUT=$R/gpurun_out/pmc2_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS=${@:-"time 30000 10000 2 0"}
i=0
for set in "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_LDS_STORE SQ_INSTS_LDS_LOAD SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- $BIN $ARGS > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'accum' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg): print(f"$TAG {k:28s} {sum(agg[k])/len(agg[k]):.6g}")
PY
find $OUT -name '*.csv' -size +1M -delete
