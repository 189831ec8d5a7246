#!/bin/bash
# One artefact that ties clock, power and kernel duration together for the SAME back-to-back
# launches of the forward kernel.  Run on the GPU box:
#   tools/power_clock.sh [out file]        -> profiles/r04_power_clock.txt (copy it there)
# Part 1 (un-profiled): the harness runs the STAMP build of k_wsplit_accum back to back
#   (150000 pairs x 10000 frames = 24 GB per launch, 250 launches): per launch the hipEvent
#   time, at the end shader cycles per unit and pass (s_memtime) and the in-kernel clock
#   = delta s_memtime / delta s_memrealtime x 100 MHz; rocm-smi samples power, its sclk and the
#   junction temperature meanwhile.
# Part 2 (rocprofv3, program directly behind --): the same launches under --pmc GRBM_GUI_ACTIVE
#   with --kernel-trace: effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/r04_power_clock.txt}
BIN=$R/tools/wfft/wfft_test
export WF_R0=${WF_R0:-20} WF_R=${WF_R:-1}
mkdir -p $(dirname $OUT); cd /tmp; export TMPDIR=/tmp
{
echo "# power_clock.sh  $(date -u +%FT%TZ)  plan R0=$WF_R0 R=$WF_R  binary sha $(sha256sum $BIN | cut -c1-16)"
echo "## part 1: un-profiled, STAMP build, 250 launches of 24 GB back to back; rocm-smi samples meanwhile"
$BIN time 150000 10000 250 1 > /tmp/pc_run.log 2>&1 &
PID=$!
sleep 1.2
for i in 1 2 3 4 5 6 7 8; do
  if kill -0 $PID 2>/dev/null; then
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor junction" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
    echo
  fi
  sleep 0.25
done
wait $PID
cat /tmp/pc_run.log
echo "## idle, 1 s later"
sleep 1
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo
echo "## part 2: rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace, 40 launches"
rm -rf /tmp/pc_prof
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pc_prof -- $BIN time 150000 10000 40 0 > /tmp/pc_prof.log 2>&1
tail -1 /tmp/pc_prof.log
python3 - <<PY
import csv, glob, statistics
dur = {}
for f in glob.glob("/tmp/pc_prof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "accum" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
cnt = {}
for f in glob.glob("/tmp/pc_prof/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "accum" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[r["Dispatch_Id"]] = float(r["Counter_Value"])
mhz = [cnt[k] / 8.0 / dur[k] for k in cnt if k in dur]
d = [dur[k] for k in cnt if k in dur]
if mhz:
    print(f"k_wsplit_accum x{len(mhz)}: duration median {statistics.median(d):.1f} us (min {min(d):.1f}, max {max(d):.1f});"
          f" GRBM_GUI_ACTIVE / 8 / duration = median {statistics.median(mhz):.0f} MHz (min {min(mhz):.0f}, max {max(mhz):.0f})")
else:
    print("no counter rows found")
PY
} > $OUT 2>&1
cat $OUT
