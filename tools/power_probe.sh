#!/bin/bash
# samples power / clocks (rocm-smi) while the forward kernel of the harness runs back to back
# usage: power_probe.sh [harness binary suffix] [harness args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
BIN=$R/tools/wfft/wfft_test; if [ -n "${1:-}" ] && [ -x "$R/tools/wfft/wfft_test_$1" ]; then BIN=$R/tools/wfft/wfft_test_$1; shift; fi
ARGS=${@:-"time 150000 10000 300 0"}
$BIN $ARGS > /tmp/power_run.log 2>&1 &
PID=$!
sleep 1.0
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr '\n' ';'
  echo
  sleep 0.4
done
wait $PID
cat /tmp/power_run.log
