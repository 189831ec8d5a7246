#!/bin/bash
# the checks the driver runs at round end, on one box: full GPU test suite, smoke, default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
python -m pytest tests -q -m gpu -x > gpurun_out/gpu_full_tests.txt 2>&1 < /dev/null; tail -4 gpurun_out/gpu_full_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/gpu_full_smoke.txt 2>&1 < /dev/null; tail -2 gpurun_out/gpu_full_smoke.txt
python bench.py > gpurun_out/gpu_full_bench.json 2> gpurun_out/gpu_full_bench.err < /dev/null; python tools/benchline.py < gpurun_out/gpu_full_bench.json; tail -2 gpurun_out/gpu_full_bench.err
