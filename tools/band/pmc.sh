#!/bin/bash
# PMC snapshot of k_band_lags from the standalone harness (run on the GPU box): issue, wait and MFMA-busy
# counters for the VACF and the Helfand form at 5000 x 50000 x 3 -> profiles/r04_band_counters.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; BIN=$R/tools/band/band_test
cd /tmp; export TMPDIR=/tmp
for mode in "" helfand; do
  OUT=/tmp/pmcb_$mode; rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- $BIN time 5000 50000 3 2 $mode > $OUT/log$i.txt 2>&1
  done
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_band_lags' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/s1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_band_lags' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("## k_band_lags ${mode:-vacf} 5000 x 50000 x 3: dur_us", [round(d/1e3,1) for d in dur])
for k in sorted(agg): print(f"{k:28s} {sum(agg[k])/len(agg[k]):.6g}")
PY
done
