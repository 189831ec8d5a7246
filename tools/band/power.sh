#!/bin/bash
# package power and sclk (rocm-smi) while the band kernels run back to back -> profiles/r04_band_power.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=$R/tools/band/band_test
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //; s/=\{10,\}//g' | tr '\n' ';'; echo; }
echo "# band/power.sh $(date -u +%FT%TZ)"
for what in "" helfand; do
  echo "## band_test time 20000 25000 3 8 $what"
  $B time 20000 25000 3 8 $what > /tmp/bp.log 2>&1 &
  PID=$!
  sleep 2.5
  for i in 1 2 3 4; do kill -0 $PID 2>/dev/null && smi; sleep 0.5; done
  wait $PID
  grep ' ms ' /tmp/bp.log | tail -2
  sleep 1
done
