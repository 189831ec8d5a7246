#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace/stats profile of bench.py plus
# separate PMC passes for HBM traffic (FETCH_SIZE and WRITE_SIZE cannot share a
# pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Outputs under gpurun_out/prof_<tag>/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
shift || true
ARGS=${@:-"--steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-host-path --no-check --no-kernel-split --no-clock-probe"}
KEY=${TA_TRAFFIC_KEY:-fft_10000x100000x3}
KERN=${TA_TRAFFIC_KERNEL:-k_wsplit_accum}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 $R/bench.py $ARGS > $OUT/bench_tcc.log 2>&1
python3 $R/tools/summarize_profile.py $OUT $KEY $KERN > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# raw traces are large; keep the summaries and the stats csv only
find $OUT -name '*kernel_trace.csv' -size +2M -delete
