#!/bin/bash
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
(timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -x -q -k "early_exit or sort" -s > gpurun_out/r04t_tests.log 2>&1; echo "exit $?" >> gpurun_out/r04t_tests.log)
tail -6 gpurun_out/r04t_tests.log | cut -c1-300
grep -q '^exit 0' gpurun_out/r04t_tests.log || exit 1
bash tools/ab.sh r04t FGOICP_CUT_TIERS "$1" "$2" 1
