#!/bin/bash
# PMC snapshot of k_band32_tp (Helfand float32, k-slots from the time axis) from the standalone harness, run on the
# GPU box: issue, wait and MFMA-busy counters at one share of configs[4], 20000 x 25000 x 3
#   -> profiles/r05_band32tp_counters.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; BIN=$R/tools/band/bandbp_test
cd /tmp; export TMPDIR=/tmp
OUT=/tmp/pmc32tp; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32" "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- $BIN ttime 20000 25000 4 > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_band32_tp' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/s1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_band32_tp' in r['Kernel_Name']:
            dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
print("## k_band32_tp 20000 x 25000 x 3 lag sums (float32 product slab 6 GB): dur_us", [round(d/1e3,1) for d in dur])
m={k:sum(v)/len(v) for k,v in agg.items()}
for k in sorted(m): print(f"{k:32s} {m[k]:.6g}")
if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'SQ_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m:
    simd_cycles = m['GRBM_GUI_ACTIVE'] / 8 * 1024   # per-XCD sum -> chip cycles x 1024 SIMDs
    print(f"matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.3f}")
if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    print(f"HBM-side bytes: 2 x FETCH_SIZE + WRITE_SIZE = {(2*m['FETCH_SIZE']+m['WRITE_SIZE'])*1024/1e9:.1f} GB (units of KB in the counters)")
if 'TCC_HIT_sum' in m: print(f"L2 hit rate {m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.3f}")
PY
