#!/bin/bash
# Round 4, session C: A/B of the item kernel on the dense and trimmed legs (development build, FGOICP_BOUNDS_ITEM = 0 / 1 alternating)
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
DEV=$PWD/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
OUT=gpurun_out/r04c_ab_item_kernel.txt
: > $OUT
for leg in dragon trimmed; do
  for item in 0 1 0 1; do
    echo "== leg $leg FGOICP_BOUNDS_ITEM=$item" | tee -a $OUT
    FGOICP_LIB=$DEV FGOICP_BOUNDS_ITEM=$item timeout -k 10 300 python3 bench.py --only $leg --steps 3 --warmup 1 2>gpurun_out/r04c_err.txt | python3 tools/bench_pick.py | tee -a $OUT
  done
done
