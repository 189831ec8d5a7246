#!/bin/bash
# timing ablations of the Helfand form of k_band_lags (wrong results by construction): no re-centring
# (ABL=1), no preparation at all (ABL=2), a new reference row every 8 / 4 steps (REF).  The variants are
# built here when missing (hipcc: ~1 min each; they travel to the GPU box once built).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/tools/band
for v in ABL=1 ABL=2 REF=8 REF=4; do
  [ -x band_test_${v/=/} ] || /opt/rocm/bin/hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=fast -DTA_BAND_$v band_test.hip -o band_test_${v/=/} 2>/dev/null
done
for v in "" _ABL1 _ABL2 _REF8 _REF4; do echo "## band_test$v"; timeout -k 10 100 $R/tools/band/band_test$v time 20000 25000 3 2 helfand | grep ms || exit 1; done
