#!/bin/bash
# HBM-side bytes (FETCH_SIZE, L2 misses incl. Infinity-Cache hits) and time of the band kernel against the
# number of octets an XCD keeps in flight and the number of XCD labels
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=$R/tools/band/band_test
cd /tmp && export TMPDIR=/tmp
for cfg in "8 0" "8 1" "8 2" "8 4" "8 16" "1 0" "4 0"; do
  set -- $cfg
  export BAND_LABELS=$1 BAND_NPH=$2
  rm -rf /tmp/bt; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/bt -- $B time ${BT_T:-5000} ${BT_A:-50000} 3 2 $BT_MODE > /tmp/bt.log 2>&1
  echo "## labels $1 n_ph $2: $(grep groups /tmp/bt.log | sed 's/.*groups/groups/')"
  grep ' ms ' /tmp/bt.log | tail -1
  python3 - <<'PY'
import csv,glob
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/bt/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(f)) if "k_band_lags" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("  FETCH_SIZE mean %.1f GB per launch (x2 corrected: %.1f GB)" % (sum(v)/len(v)*1024/1e9, 2*sum(v)/len(v)*1024/1e9))
PY
done
