#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
python -m pytest tests/test_integration_snippet.py tests/test_gpu_dist.py tests/test_cabi_and_dist.py -q -m gpu -x > gpurun_out/r4_run5_a.txt 2>&1; tail -15 gpurun_out/r4_run5_a.txt
python -m pytest tests/test_api.py -q -m gpu -x > gpurun_out/r4_run5_b.txt 2>&1; tail -15 gpurun_out/r4_run5_b.txt
