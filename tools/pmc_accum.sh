#!/bin/bash
# PMC snapshot of the FFT accumulate kernel (run on the GPU box).  usage: pmc_accum.sh TAG "probe args"
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; ARGS=${2:-"--cases 10000x20000 --reps 2"}
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- python3 $R/tools/gpu_probe.py $ARGS > /dev/null 2>&1
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
print("accumulate kernel dur_us", [d/1e3 for d in dur])
for k in sorted(agg): print(f"{k:28s} {sum(agg[k])/len(agg[k]):.5g}")
PY
rm -rf $OUT
