#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
( cd tools/wfft
echo "== R0=20"; ./wfft_test time 150000 10000 5 1
echo "== R0=18"; WF_R0=18 ./wfft_test time 160000 9216 5 1
echo "== R0=16"; WF_R0=16 ./wfft_test time 180000 8192 5 1
echo "== R0=14"; WF_R0=14 ./wfft_test time 200000 7168 5 1
echo "== R0=12"; WF_R0=12 ./wfft_test time 240000 6144 5 1
echo "== R0=20 without S2 (timing ablation)"; ./wfft_test_abl1 time 150000 10000 5 1
echo "== R0=20 without row loads (timing ablation, spills 292 B)"; ./wfft_test_abl3 time 150000 10000 5 1
) > gpurun_out/r4_run3.txt 2>&1
cat gpurun_out/r4_run3.txt
python -m pytest tests/test_integration_snippet.py -q -m gpu -x 2>&1 | tail -3
