#!/bin/bash
# joules per FP64 FMA, per LDS byte, per HBM byte: package power (rocm-smi) while each micro-kernel
# runs back to back for ~4 s -> profiles/r04_energy_ubench.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=$R/tools/ubench/energy_bench
smi() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //; s/=\{10,\}//g' | tr '\n' ';'; echo; }
echo "# energy_bench.sh $(date -u +%FT%TZ)"
echo "## idle"; sleep 1; smi
for what in ${EB_KERNELS:-fma lds copy mfma mfma4}; do
  echo "## $what"
  $B $what 4 > /tmp/eb_$what.log 2>&1 &
  PID=$!
  sleep 1.5
  for i in 1 2 3 4; do kill -0 $PID 2>/dev/null && smi; sleep 0.4; done
  wait $PID
  cat /tmp/eb_$what.log
  sleep 1
done
