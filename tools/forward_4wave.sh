#!/bin/bash
# the headline plan (R0 = 20, 10000 frames) on 4 waves x 512 registers, same box, old vs new:
#   wfft_test        the library's kernel: 8 waves per workgroup (two per SIMD), 20 sub-series as 3 + 2 per wave,
#                    the next unit's 20 row requests in one burst behind S2
#   wfft_test_nw4    -DWF_NW_R0=20 -DWF_NW_VAL=4: one wave per SIMD, 5 sub-series per wave, two first-stage
#                    butterflies per thread (40 rows), the burst behind S2
#   wfft_test_nw4s   ... -DWF_SPREAD_ALL=1: the next unit's 40 rows requested along S2 (the compiler parks them
#                    in AGPRs: 152 of them)
# per binary: check, then 250 launches back to back (24 GB each) with rocm-smi sampling package power and sclk,
# the harness printing hipEvent time, cycles per unit and pass and the in-kernel clock.
# Run on the GPU box: tools/forward_4wave.sh -> profiles/r05_forward_4wave.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/r05_forward_4wave.txt}
mkdir -p $(dirname $OUT); cd /tmp
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //; s/=\{10,\}//g' | tr '\n' ';'; echo; }
{
echo "# forward_4wave.sh $(date -u +%FT%TZ)"
for v in "" _nw4 _nw4s; do
  BIN=$R/tools/wfft/wfft_test$v
  echo "## wfft_test$v  (sha $(sha256sum $BIN | cut -c1-16))"
  WF_R0=20 timeout -k 10 100 $BIN check 2>&1 | tail -2 | head -1
  WF_R0=20 $BIN time 150000 10000 250 1 > /tmp/f4.log 2>&1 &
  PID=$!
  sleep 1.2
  for i in 1 2 3 4; do kill -0 $PID 2>/dev/null && smi; sleep 0.25; done
  wait $PID
  tail -3 /tmp/f4.log
done
} 2>&1 | tee $OUT
