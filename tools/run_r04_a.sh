#!/bin/bash
# Round 4, session A: (1) VALU issue rates of gfx950 (plain vs packed fp32), (2) the command whose teardown aborted in round 3, with
# its stderr kept and its status checked, (3) the multi-rank GPU tests on the refactored core, (4) a bench line on this box.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 120 tools/calib/valu_rate > gpurun_out/r04a_valu_rate.txt 2>&1; echo "valu_rate exit $?" | tee -a gpurun_out/r04a_valu_rate.txt
tail -14 gpurun_out/r04a_valu_rate.txt
tools/scale_replays.sh r04a "FGOICP_REPLAY_SCHEDULE=serial 8 dragon 5e-6 0.005 1" || exit 1
(timeout -k 10 900 python3 -m pytest tests/test_gpu_multi.py -x -q > gpurun_out/r04a_gputests_multi.log 2>&1; echo "exit $?" >> gpurun_out/r04a_gputests_multi.log)
tail -3 gpurun_out/r04a_gputests_multi.log
grep -q '^exit 0' gpurun_out/r04a_gputests_multi.log || exit 1
(timeout -k 10 600 python3 bench.py > gpurun_out/r04a_bench_default.log 2>&1; echo "exit $?" >> gpurun_out/r04a_bench_default.log)
tail -2 gpurun_out/r04a_bench_default.log | cut -c1-600
