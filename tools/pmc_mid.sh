#!/bin/bash
# PMC snapshot of k_mid (run on the GPU box): pmc_mid.sh TAG T  (12 GB of input, lag sums, "mid_max" 512)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; T=${2:-128}
OUT=$R/gpurun_out/pmcm_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/pmc_mid_run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import torch, bench
from transport_analysis_amd import _lib
ctx = _lib.Context(0); ctx.set_option("mid_max", 512)
T = $T; A = int(5e8 / T) // 64 * 64
c = bench.Case(torch, ctx, torch.device("cuda:0"), "direct", T, A, 3, 0, A * 3, bench.SEED + 4, False, False, False, False)
for r in range(3):
    c.step(); torch.cuda.synchronize()
print("ms", ctx.last_timing())
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- python3 /tmp/pmc_mid_run.py > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_mid' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/s1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_mid' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("$TAG k_mid T=$T dur_us", [round(d/1e3,1) for d in dur])
for k in sorted(agg): print(f"{k:28s} {sum(agg[k])/len(agg[k]):.6g}")
PY
find $OUT -name '*.csv' -size +1M -delete
