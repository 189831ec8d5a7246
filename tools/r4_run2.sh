#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
python tools/bp_sweep.py 10000 100000 3 > gpurun_out/r4_bp_sweep.txt 2>&1
python tools/bp_sweep.py 20000 25000 3 0,2048,1024,512,256 >> gpurun_out/r4_bp_sweep.txt 2>&1
cat gpurun_out/r4_bp_sweep.txt
( cd tools/wfft; echo "== R0=8 4096 frames: 2 WG/CU (as built) / 1 WG/CU";  WF_R0=8 ./wfft_test time 360000 4096 5 1; WF_PERCU=1 WF_R0=8 ./wfft_test time 360000 4096 5 1 ) > gpurun_out/r4_run2.txt 2>&1
cat gpurun_out/r4_run2.txt
bash tools/power_clock.sh $R/gpurun_out/r04_power_clock.txt > /dev/null 2>&1
python -m pytest tests/test_integration_snippet.py tests/test_gpu_dist.py tests/test_cabi_and_dist.py -q -m gpu -x > gpurun_out/r4_run2_tests.txt 2>&1
tail -5 gpurun_out/r4_run2_tests.txt
