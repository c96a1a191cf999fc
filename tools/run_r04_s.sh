#!/bin/bash
# same-box A/B of two builds of the library (FGOICP_LIB): the previous commit's against this tree's, exact mode and early exit
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04s_ab_builds.txt
: > $OUT
for rep in 1 2; do
for lib in libfgoicp_amd_prev.so libfgoicp_amd.so; do
  for flag in "--no-early-exit" ""; do
    for leg in "$@"; do
      echo "== $lib leg $leg $flag" | tee -a $OUT
      FGOICP_LIB=$PWD/fast-go-icp_amd/lib/$lib timeout -k 10 400 python3 bench.py --only $leg $flag --no-full-evaluation 2> gpurun_out/r04s.err | python3 tools/bench_pick.py | tee -a $OUT
      [ "${PIPESTATUS[0]}" -eq 0 ] || { tail -20 gpurun_out/r04s.err; exit 1; }
    done
  done
done
done
