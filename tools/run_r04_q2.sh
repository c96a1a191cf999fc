#!/bin/bash
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
export FGOICP_LIB=$PWD/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
WL=${1:-bunny}; G=${2:-256}
OUT=gpurun_out/r04q2_cut_cost_$WL.txt
: > $OUT
run() { echo "== $*" | tee -a $OUT; env "$@" timeout -k 10 300 python3 tools/op_bench.py $WL $G 5 2>>gpurun_out/r04q.err | tee -a $OUT; }
run A=exact
run OP_BENCH_CUT=2
run OP_BENCH_CUT=0.5
run OP_BENCH_CUT=0.1
run OP_BENCH_CUT=0.0
