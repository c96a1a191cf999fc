#!/bin/bash
# A/B of a development-build knob on bench legs, alternating arms, stderr kept:   tools/ab.sh TAG KNOB "VAL_A VAL_B ..." "LEG ..." [REPEATS]
# e.g.  tools/ab.sh r04_item FGOICP_BOUNDS_ITEM "0 1" "dragon trimmed" 2
# Replaces round 3's one-off ab_*.sh scripts (their records are under profiles/r03_ab_*.txt; the knobs they flip exist in the development
# build only: libfgoicp_amd_dev.so, csrc/host/knobs.hpp).
set -u -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=$1; KNOB=$2; VALS=$3; LEGS=$4; REP=${5:-2}
mkdir -p gpurun_out
export FGOICP_LIB=${FGOICP_LIB:-$PWD/fast-go-icp_amd/lib/libfgoicp_amd_dev.so}
OUT=gpurun_out/${TAG}_ab.txt
: > "$OUT"
for leg in $LEGS; do
  for r in $(seq 1 "$REP"); do
    for v in $VALS; do
      echo "== leg $leg $KNOB=$v (repeat $r)" | tee -a "$OUT"
      env "$KNOB=$v" timeout -k 10 300 python3 bench.py --only "$leg" --steps 3 --warmup 1 2> "gpurun_out/${TAG}_ab_${leg}_${v}_${r}.err" | python3 tools/bench_pick.py | tee -a "$OUT"
      rc=${PIPESTATUS[0]}
      [ "$rc" -eq 0 ] || { echo "bench.py exited with status $rc; stderr:"; tail -20 "gpurun_out/${TAG}_ab_${leg}_${v}_${r}.err"; exit "$rc"; }
    done
  done
done
