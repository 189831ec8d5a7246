#!/bin/bash
# PMC snapshot of k_wsplit_accum from the standalone harness (run on the GPU box).
# usage: pmc.sh TAG [BINARY-SUFFIX] [harness args...]   (environment: WF_R0, WF_R as for the harness)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; shift
BIN=$R/tools/wfft/wfft_test; if [ -n "$1" ] && [ -x "$R/tools/wfft/wfft_test_$1" ]; then BIN=$R/tools/wfft/wfft_test_$1; shift; fi
OUT=$R/gpurun_out/pmcw_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
ARGS=${@:-"time 30000 10000 2 0"}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- $BIN $ARGS > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'accum' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/s1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'accum' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("$TAG accum dur_us", [round(d/1e3,1) for d in dur])
for k in sorted(agg): print(f"{k:28s} {sum(agg[k])/len(agg[k]):.6g}")
PY
find $OUT -name '*.csv' -size +1M -delete
