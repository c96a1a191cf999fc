#!/bin/bash
# final tree: extended sort-fault test (dev child), then the randomised campaigns with the threshold cases in
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
(timeout -k 10 600 python3 -m pytest tests/test_gpu_dev_build.py tests/test_gpu_fullsize.py -m gpu -x -q -k "dev_build or early_exit" > gpurun_out/r04w_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04w_tests.log)
tail -4 gpurun_out/r04w_tests.log | cut -c1-300
grep -q '^exit 0' gpurun_out/r04w_tests.log || exit 1
timeout -k 10 500 python3 tools/fuzz_gpu.py 200 51 > gpurun_out/r04w_fuzz_ops.txt 2>&1; echo "ops exit $?"; tail -2 gpurun_out/r04w_fuzz_ops.txt
timeout -k 10 500 python3 tools/fuzz_gpu.py 60 53 run > gpurun_out/r04w_fuzz_runs.txt 2>&1; echo "runs exit $?"; tail -2 gpurun_out/r04w_fuzz_runs.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
