#!/bin/bash
# early exit on / off on the other legs
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04r_ab_early_exit.txt
: > $OUT
for leg in "$@"; do
  for flag in "--no-early-exit" ""; do
    echo "== leg $leg $flag" | tee -a $OUT
    timeout -k 10 400 python3 bench.py --only $leg $flag 2> gpurun_out/r04r_${leg}.err | python3 tools/bench_pick.py | tee -a $OUT
    [ "${PIPESTATUS[0]}" -eq 0 ] || { tail -20 gpurun_out/r04r_${leg}.err; exit 1; }
  done
done
