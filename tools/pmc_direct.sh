#!/bin/bash
# PMC snapshot of the direct correlator (run on the GPU box).
# usage: pmc_direct.sh TAG [bench.py workload args]; default BASELINE configs[3];
#   configs[4] share, float32 path: pmc_direct.sh f32 --mode helfand --float32 --frames 20000 --atoms 25000
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-x}; shift || true
OUT=$R/gpurun_out/pmcd_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
WL=${@:-"--mode direct --frames 5000 --atoms 50000"}
ARGS="$WL --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-host-path --no-check --no-kernel-split"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM_RD" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- python3 $R/bench.py $ARGS > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_direct' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/s1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_direct' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("$TAG k_direct dur_us", [round(d/1e3,1) for d in dur])
for k in sorted(agg): print(f"{k:28s} {sum(agg[k])/len(agg[k]):.6g}")
PY
find $OUT -name '*.csv' -size +1M -delete
