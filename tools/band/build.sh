#!/bin/bash
# builds tools/band/band_test and prints the kernel's register budget
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" band_test.hip -o band_test 2>&1 | python3 ../wfft/kres.py
