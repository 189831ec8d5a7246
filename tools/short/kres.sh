#!/bin/bash
# register budget of one variant: kres.sh [-DTM=.. -DMD=.. -DDM=..]
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-function -I../../transport_analysis_amd/csrc "$@" \
  -Rpass-analysis=kernel-resource-usage short_test.hip -o /tmp/short_kres 2>&1 | grep -E "Function Name: _ZN2ta7k_short|VGPRs:|ScratchSize|Occupancy" | paste - - - - | sed 's/remark: [^ ]* //g; s/short_test.hip:[0-9:]* //g'
