#!/bin/bash
# builds tools/band/bandbp_test[_SUFFIX] and prints the kernel's register budget
#   buildbp.sh [SUFFIX] [-DBP_NW=8] ...
cd "$(dirname "$0")"
SUF=""
if [ $# -gt 0 ] && [ "${1#-}" = "$1" ]; then SUF="_$1"; shift; fi
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" bandbp_test.hip -o bandbp_test$SUF 2>&1 | python3 ../wfft/kres.py
