#!/bin/bash
# builds tools/wfft/wfft_test[_SUFFIX] and prints the register budget of the wfft kernels
# usage: build.sh [-o suffix] [extra hipcc flags, e.g. -DWF_ABL=2]
cd "$(dirname "$0")"
OUT=wfft_test
if [ "$1" = "-o" ]; then OUT=wfft_test_$2; shift 2; fi
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" wfft_test.hip -o $OUT 2>&1 | python3 kres.py
