#!/bin/bash
# Round 4, session E: the dense leg is bound by the L1's tag rate (0.94 cache-line accesses per clock and CU, TA busy 79 %): which packed layout
# needs the fewest line accesses per point-evaluation?  z-pair (default for dense clouds), yz-quad runs, apron quads — time and TCP counters.
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/profiles
export TMPDIR=/tmp
DEV=$REPO/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
OUT=$REPO/gpurun_out/r04e_ab_dense_layouts.txt
: > $OUT
cd /tmp
SET4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"
for layout in 1 2 4 1 2 4; do
  echo "== dragon leg, FGOICP_LUT_ZPAIR=$layout" | tee -a $OUT
  FGOICP_LIB=$DEV FGOICP_LUT_ZPAIR=$layout timeout -k 10 300 python3 $REPO/bench.py --only dragon --steps 1 --warmup 0 2>gpurun_out/r04b_err.txt | python3 $REPO/tools/bench_pick.py | tee -a $OUT
done
for layout in 2 4; do
  rm -rf /tmp/p4_l$layout
  FGOICP_LIB=$DEV FGOICP_LUT_ZPAIR=$layout timeout -k 10 300 rocprofv3 --pmc $SET4 --kernel-trace --output-format csv -d /tmp/p4_l$layout -- python3 $REPO/bench.py --only dragon --steps 1 --warmup 0 > $REPO/gpurun_out/r04e_px4_layout$layout.log 2>&1 || { echo "TCP pass layout=$layout failed"; exit 1; }
  echo "== TCP counters, layout $layout" | tee -a $OUT
  python3 $REPO/tools/pmc_generic.py $REPO/gpurun_out/profiles/r04_dragon_layout${layout}_pmc_tcp.json /tmp/p4_l$layout | grep -E "^bounds_(sorted|item)_kernel" | tee -a $OUT
done
