#!/bin/bash
# Round 4, session B: the item kernel (bounds_item.hpp) against round 3's family — bits first (development build), then time per leg.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
DEV=$PWD/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
(FGOICP_LIB=$DEV timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -k "item_kernel or packed_lut or twin or bounds" > gpurun_out/r04b_gputests_item.log 2>&1; echo "exit $?" >> gpurun_out/r04b_gputests_item.log)
tail -4 gpurun_out/r04b_gputests_item.log
grep -q '^exit 0' gpurun_out/r04b_gputests_item.log || exit 1
OUT=gpurun_out/r04b_ab_item_kernel.txt
: > $OUT
for leg in dragon headline trimmed; do
  for item in 0 1 0 1; do
    echo "== leg $leg FGOICP_BOUNDS_ITEM=$item" | tee -a $OUT
    FGOICP_LIB=$DEV FGOICP_BOUNDS_ITEM=$item timeout -k 10 300 python3 bench.py --only $leg --steps 3 --warmup 1 2>gpurun_out/r04b_err.txt | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        r = d.get('roofline', {})
        print(json.dumps({k: d.get(k) for k in ('value', 'ms_per_step')} | {k: r.get(k) for k in ('avg_launch_us', 'launches', 'evaluations_per_launch', 'bound', 'achieved', 'frac')}))
" | tee -a $OUT
  done
done
