#!/bin/bash
# L2 hit rate + fabric bytes + time of a harness command: pmc_l2.sh KERNEL_SUBSTR BIN args...
K=$1; shift
cd /tmp; export TMPDIR=/tmp
OUT=/tmp/pmcl2; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -- "$@" > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if '$K' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/s1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if '$K' in r['Kernel_Name']: dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
m={k:sum(v)/len(v) for k,v in agg.items()}
print("$K", "dur_ms", [round(d/1e6,2) for d in dur], "fabric GB %.1f" % ((2*m.get('FETCH_SIZE',0)+m.get('WRITE_SIZE',0))*1024/1e9), "L2 hit %.3f" % (m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum'])))
PY
