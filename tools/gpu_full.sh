#!/bin/bash
# the checks the driver runs at round end, on one box: full GPU test suite, smoke, default bench.
# Exit status = the first failing step's.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export PYTHONPATH=$R
mkdir -p gpurun_out
rc=0
python -m pytest tests -q -m gpu -x > gpurun_out/gpu_full_tests.txt 2>&1 < /dev/null || rc=$?
tail -4 gpurun_out/gpu_full_tests.txt
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/gpu_full_smoke.txt 2>&1 < /dev/null || rc=$?
tail -2 gpurun_out/gpu_full_smoke.txt
[ $rc -ne 0 ] && exit $rc
python bench.py --full-json gpurun_out/gpu_full_bench_full.json > gpurun_out/gpu_full_bench.json 2> gpurun_out/gpu_full_bench.err < /dev/null || rc=$?
python tools/benchline.py < gpurun_out/gpu_full_bench.json; tail -2 gpurun_out/gpu_full_bench.err
exit $rc
