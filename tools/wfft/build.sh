#!/bin/bash
# builds tools/wfft/wfft_test[_SUFFIX] and prints the register budget of the wfft kernels
# usage: build.sh [-o suffix] [extra hipcc flags, e.g. -DWF_ABL=2]
cd "$(dirname "$0")"
OUT=wfft_test
if [ "$1" = "-o" ]; then OUT=wfft_test_$2; shift 2; fi
# -DWF_ABL=n / -DWF_INV_STAMP=1 builds: the instrumentation lives in ablations.patch since round 6
case "$*" in *WF_ABL*|*WF_INV_STAMP*)
  patch -s -o wfft_abl.hpp ../../transport_analysis_amd/csrc/wfft.hpp ablations.patch || exit 1
  sed -i 's|#include "fft_engine.hpp"|#include "../../transport_analysis_amd/csrc/fft_engine.hpp"|; s|#include "wfft_twist.inc"|#include "../../transport_analysis_amd/csrc/wfft_twist.inc"|' wfft_abl.hpp
  set -- -DWF_ABLATIONS "$@";;
esac
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage "$@" wfft_test.hip -o $OUT 2>&1 | python3 kres.py
