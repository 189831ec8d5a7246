#!/bin/bash
# builds tools/wfft/wfft_test and prints the register budget of the accumulate kernels
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" wfft_test.hip -o wfft_test 2>&1 | \
  grep -E "error|Function Name|VGPRs:|Scratch" | grep -A2 "k_w" | sed -e 's/.*remark: *//' | paste - - - | sed -e 's/\[-Rpass[^]]*\]//g'
