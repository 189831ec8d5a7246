#!/bin/bash
# One artefact that ties clock, power and kernel duration together for the SAME back-to-back
# launches of the forward kernel, and the energy per input byte of every plan.
# Run on the GPU box:   tools/power_clock.sh [out file]   -> copy to profiles/r04_power_clock.txt
# Part 1 (un-profiled): per plan the harness runs the STAMP build of k_wsplit_accum back to back
#   (~24 GB per launch, ~2.5 s): hipEvent time per launch, shader cycles per unit and pass
#   (s_memtime), in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz; rocm-smi
#   samples package power, its sclk and the junction temperature meanwhile.  Then the same binary
#   on ALL-ZERO input (WF_ZERO=1): the same instruction stream at a fraction of the power.
# Part 2 (rocprofv3, program directly behind --): R0 = 20 under --pmc GRBM_GUI_ACTIVE with
#   --kernel-trace: effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/r04_power_clock.txt}
BIN=$R/tools/wfft/wfft_test
mkdir -p $(dirname $OUT); cd /tmp; export TMPDIR=/tmp
smi() { rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor junction" | sed 's/^GPU\[0\]\s*: //; s/=\{10,\}//g' | tr '\n' ';'; echo; }
plan() {  # R0 pairs T launches [zero]
  export WF_R0=$1 WF_R=1 WF_ZERO=${5:-0}
  echo "### plan R0=$1 ($3 frames x $2 pairs, $4 launches back to back)  zero_input=$WF_ZERO"
  $BIN time $2 $3 $4 1 > /tmp/pc_run.log 2>&1 &
  local PID=$!
  sleep 1.3
  for i in 1 2 3 4; do kill -0 $PID 2>/dev/null && smi; sleep 0.25; done
  wait $PID
  cat /tmp/pc_run.log
}
{
echo "# power_clock.sh  $(date -u +%FT%TZ)  harness sha $(sha256sum $BIN | cut -c1-16)"
echo "## part 1: un-profiled STAMP build; rocm-smi samples while the launches run"
plan 20 150000 10000 250
plan 16 180000 8192 280
plan 12 240000 6144 290
plan 10 300000 5120 270
plan 8 360000 4096 280
echo "## the same on all-zero input"
plan 20 150000 10000 250 1
plan 12 240000 6144 290 1
plan 8 360000 4096 280 1
echo "## idle, 1 s later"; sleep 1; smi
echo "## part 2: rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace, R0 = 20, 40 launches"
export WF_R0=20 WF_R=1 WF_ZERO=0
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
