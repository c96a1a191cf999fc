#!/bin/bash
# Folding the tick sort's permutation check into bounds_item_kernel: parity of the tests that exercise it, then an A/B against the separate launch.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
(timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ops.py tests/test_gpu_dev_build.py -m gpu -x -q > gpurun_out/r04o_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04o_tests.log)
tail -4 gpurun_out/r04o_tests.log | cut -c1-300
grep -q '^exit 0' gpurun_out/r04o_tests.log || exit 1
bash tools/ab.sh r04o FGOICP_SEPARATE_CHECK "1 0" "default_threshold serial headline" 3
