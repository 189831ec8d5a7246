#!/bin/bash
# builds tools/band/band32_test[_SUFFIX] and prints the kernel's register budget
#   build32.sh [SUFFIX] [-DB32_NW=4|8] [-DB32_P=1|2|4|8] ...
cd "$(dirname "$0")"
SUF=""
if [ $# -gt 0 ] && [ "${1#-}" = "$1" ]; then SUF="_$1"; shift; fi
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" band32_test.hip -o band32_test$SUF 2>&1 | python3 ../wfft/kres.py
