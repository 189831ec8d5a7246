#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" _ABL1 _ABL2 _REF8 _REF16; do echo "## band_test$v"; timeout -k 10 100 $R/tools/band/band_test$v time 20000 25000 3 2 helfand | grep ms || exit 1; done
