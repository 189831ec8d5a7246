#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
python -m pytest tests/test_integration_snippet.py -q -m gpu -x > gpurun_out/r4_run4_snip.txt 2>&1; tail -30 gpurun_out/r4_run4_snip.txt
python -m pytest tests/test_gpu_parity.py -q -x -k "fft or particle or golden or kat" > gpurun_out/r4_run4_tests.txt 2>&1; tail -5 gpurun_out/r4_run4_tests.txt
python tools/bp_sweep.py 10000 100000 3 0 > gpurun_out/r4_run4_bp.txt 2>&1
python tools/bp_sweep.py 20000 25000 3 0 >> gpurun_out/r4_run4_bp.txt 2>&1
python tools/bp_sweep.py 10000 100000 1 0 >> gpurun_out/r4_run4_bp.txt 2>&1
python tools/bp_sweep.py 4000 250000 3 0 >> gpurun_out/r4_run4_bp.txt 2>&1
cat gpurun_out/r4_run4_bp.txt
