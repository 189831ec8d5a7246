#!/bin/bash
# VERDICT r05 item 2 (headline, one variant): the next unit's rows TOUCHED into the L2 during S2 by the waves that own two
# sub-series (tools/wfft/l2_touch.patch on csrc/wfft.hpp), against the library's kernel, same box: 250 launches back to back
# (24 GB each), rocm-smi sampling package power and sclk, in-kernel stamps (cycles per unit and pass, clock).
#   wfft_test_t0 = -DWF_TOUCH=0 (the library's code), wfft_test_t1 = -DWF_TOUCH=1, wfft_test_t3 = touches with sc0 (L1 bypass)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/r06_headline_touch_raw.txt}
mkdir -p $(dirname $OUT); cd /tmp
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //; s/=\{10,\}//g' | tr '\n' ';'; echo; }
{
echo "# touch_report.sh $(date -u +%FT%TZ)"
for v in t0 t1 t3 t0; do
  BIN=$R/tools/wfft/wfft_test_$v
  echo "## wfft_test_$v  (sha $(sha256sum $BIN | cut -c1-16))"
  WF_R0=20 timeout -k 10 100 $BIN check 2>&1 | tail -2 | head -1
  WF_R0=20 $BIN time 150000 10000 250 1 > /tmp/f4.log 2>&1 &
  PID=$!
  sleep 1.2
  for i in 1 2 3 4; do kill -0 $PID 2>/dev/null && smi; sleep 0.25; done
  wait $PID
  tail -3 /tmp/f4.log
done
} 2>&1 | tee $OUT
