#!/bin/bash
# on the GPU box: correctness of the band kernels, then their rates at configs[3] and configs[4]'s share
# usage: run.sh [helfand]
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=$R/tools/band/band_test
echo "# band_test $(date -u +%FT%TZ) $1"
timeout -k 10 120 $B check $1 || exit 1
timeout -k 10 120 $B time 5000 50000 3 5 $1 || exit 1
timeout -k 10 120 $B time 20000 25000 3 3 $1 || exit 1
timeout -k 10 60 $B time 1000 100000 3 3 $1 || exit 1
