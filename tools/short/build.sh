#!/bin/bash
# builds tools/short/short_test_SUFFIX: build.sh SUFFIX [hipcc flags: -DTM=64 -DMD=0 -DSHORT_ABL=3 ...]
# (-DSHORT_ABL builds take the kernel header from ablations.patch)
cd "$(dirname "$0")"
OUT=short_test_$1; shift
INC=../../transport_analysis_amd/csrc
case "$*" in *SHORT_ABL*)
  mkdir -p abl && patch -s -o abl/short_kernels.hpp $INC/short_kernels.hpp ablations.patch || exit 1
  sed -i 's|#include "direct_kernels.hpp"|#include "../../../transport_analysis_amd/csrc/direct_kernels.hpp"|' abl/short_kernels.hpp
  INC=abl;;
esac
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function -I$INC "$@" short_test.hip -o $OUT
