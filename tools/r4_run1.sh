#!/bin/bash
# round 4, GPU run 1: occupancy variants of the forward kernel, by-particle block sweep, power/clock
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R/tools/wfft; O=$R/gpurun_out/r4_run1.txt
{
W=./wfft_test
$W check; WF_R0=10 WF_R=2 $W check; WF_R0=10 WF_R=2 ./wfft_test_mw4 check; WF_R0=10 ./wfft_test_mw3 check
echo "== base R0=20"; $W time 150000 10000 5 1
echo "== R0=10 R=2 at 10000 frames (library LONG kernel, 1 WG/CU)"; WF_R0=10 WF_R=2 $W time 150000 10000 5 1
echo "== R0=10 R=2 mw4 (128 VGPR, spills, 2 WG/CU)"; WF_R0=10 WF_R=2 ./wfft_test_mw4 time 150000 10000 5 1
echo "== R0=10 R=2 mw3"; WF_R0=10 WF_R=2 ./wfft_test_mw3 time 150000 10000 5 1
echo "== R0=10 R=1 5120 frames: base / mw4 / mw3"
WF_R0=10 $W time 300000 5120 5 1; WF_R0=10 ./wfft_test_mw4 time 300000 5120 5 1; WF_R0=10 ./wfft_test_mw3 time 300000 5120 5 1
echo "== R0=8 4096 frames: 2 WG/CU (as built) / 1 WG/CU"
WF_R0=8 $W time 360000 4096 5 1; WF_PERCU=1 WF_R0=8 $W time 360000 4096 5 1
echo "== R0=12 6144 frames"; WF_R0=12 $W time 240000 6144 5 1
echo "== R0=16 8192 frames"; WF_R0=16 $W time 180000 8192 5 1
} > $O 2>&1
cd $R
python tools/bp_sweep.py 10000 100000 3 > gpurun_out/r4_bp_sweep.txt 2>&1
python tools/bp_sweep.py 20000 25000 3 0,2048,1024,512,256 >> gpurun_out/r4_bp_sweep.txt 2>&1
bash tools/power_clock.sh $R/gpurun_out/r04_power_clock.txt > /dev/null 2>&1
tail -5 $O; tail -12 gpurun_out/r4_bp_sweep.txt
