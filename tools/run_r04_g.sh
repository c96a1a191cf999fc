#!/bin/bash
# Round 4, session G: a round's refinements next to its tasks (FGOICP_OVERLAP_ICP = 0 / 1, development build), leg by leg; then the multi + fullsize tests.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
DEV=$PWD/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
OUT=gpurun_out/r04g_ab_overlap_icp.txt
: > $OUT
for leg in default_threshold headline trimmed dragon; do
  for ov in 0 1 0 1; do
    echo "== leg $leg FGOICP_OVERLAP_ICP=$ov" | tee -a $OUT
    FGOICP_LIB=$DEV FGOICP_OVERLAP_ICP=$ov timeout -k 10 300 python3 bench.py --only $leg --steps 5 --warmup 1 2>gpurun_out/r04g_err.txt | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        def find(o, key):
            if isinstance(o, dict):
                if key in o: yield o[key]
                for v in o.values(): yield from find(v, key)
        print('wall_clock_to_optimum_s', [round(x, 5) for x in find(d, 'wall_clock_to_optimum_s')], 'seconds_icp', [round(x, 5) for x in find(d, 'seconds_icp_rank0')] or [round(x,5) for x in find(d, 'seconds_icp')], 'best_sse', list(find(d, 'best_sse'))[:2], 'subcubes_per_step', list(find(d, 'subcubes_per_step'))[:2])
" | tee -a $OUT
  done
done
(timeout -k 10 900 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r04g_gputests.log 2>&1; echo "exit $?" >> gpurun_out/r04g_gputests.log)
tail -3 gpurun_out/r04g_gputests.log
