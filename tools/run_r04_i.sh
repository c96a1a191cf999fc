#!/bin/bash
# Round 4, session I: the bench line of this tree, then the N-rank replays (stderr kept per run, python's exit status).
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/profiles
(timeout -k 10 600 python3 bench.py > gpurun_out/r04i_bench_default.log 2>&1; echo "exit $?" >> gpurun_out/r04i_bench_default.log)
grep '^{"metric"' gpurun_out/r04i_bench_default.log > gpurun_out/profiles/r04_bench_line.json
tail -2 gpurun_out/r04i_bench_default.log | cut -c1-300
grep -q '^exit 0' gpurun_out/r04i_bench_default.log || exit 1
tools/scale_replays.sh r04i \
  "X=0 8 bunny 5e-5 0.005 2" \
  "X=0 8 dragon 5e-6 0.005 1" \
  "X=0 4 dragon 5e-6 0.005 1" \
  "X=0 2 dragon 5e-6 0.005 1" \
  "X=0 4 bunny 5e-5 0.005 2" \
  "X=0 2 bunny 5e-5 0.005 2" \
  "FGOICP_REPLAY_SCHEDULE=serial 8 bunny 5e-5 0.005 2" \
  "FGOICP_REPLAY_SCHEDULE=serial 8 dragon 5e-6 0.005 1" \
  "FGOICP_REPLAY_TRIM=0.2 8 synthetic1m_outliers 1e-3 0.005 0"
