#!/bin/bash
# Same-box A/B of two builds of libta_hip.so (boxes of the pool differ by a few percent, so
# numbers from different gpurun calls do not compare).  Usage, from the repo root:
#   cp transport_analysis_amd/libta_hip.so transport_analysis_amd/libta_A.so     # build A
#   ... change sources, make ...
#   cp transport_analysis_amd/libta_hip.so transport_analysis_amd/libta_B.so     # build B
#   gpurun -- 'bash tools/ab_bench.sh [bench.py args]'
# Runs bench.py twice per build, alternating, and prints the dominant kernel's time.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R/transport_analysis_amd" || exit 1
for r in 1 2; do
  for v in A B; do
    cp "libta_$v.so" libta_hip.so || exit 1
    printf '%s: ' "$v"
    (cd "$R" && python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("kernel %.3f ms, step %.3f ms" % (d["roofline"]["kernel_ms"], d["ms_per_step"]))')
  done
done
cp libta_B.so libta_hip.so
