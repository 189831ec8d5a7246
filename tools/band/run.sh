#!/bin/bash
# on the GPU box: correctness of the band kernel, then its rate at configs[3] and configs[4]'s share
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=$R/tools/band/band_test
echo "# band_test $(date -u +%FT%TZ)"
timeout -k 10 120 $B check || exit 1
timeout -k 10 120 $B time 5000 50000 3 5 || exit 1
timeout -k 10 120 $B time 20000 25000 3 3 || exit 1
timeout -k 10 60 $B time 1000 100000 3 3 || exit 1
