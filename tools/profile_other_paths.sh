set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r01k_other
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/byp -- python3 $R/bench.py --by-particle --atoms 20000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/byp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/long -- python3 $R/bench.py --frames 20000 --atoms 20000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/long.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/long50k -- python3 $R/bench.py --frames 50000 --atoms 4000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/long50k.log 2>&1
for d in byp long long50k; do echo "== $d"; f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); python3 - "$f" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if n.startswith('k_') or 'ta::' in n or 'k_' in n[:40]:
        print(f"{n[:70]:70s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):12.0f} pct={r['Percentage']}")
PY
done
find $OUT -name '*kernel_trace.csv' -delete
