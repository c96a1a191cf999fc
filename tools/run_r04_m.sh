#!/bin/bash
# Round 4, session M: the exit-time abort, A/B of the RCCL loading (development build): round 3's way (system librccl.so.1, RTLD_GLOBAL) against
# this round's (the process's own librccl if it has one, else RTLD_LOCAL | RTLD_DEEPBIND), torch first and RCCL first.  Status of every child kept.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
OUT=gpurun_out/r04m_exit_abort_ab.txt
: > $OUT
DEV=$PWD/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
for legacy in 1 0; do
  for order in torch_first rccl_first; do
    echo "== FGOICP_RCCL_LOAD_GLOBAL=$legacy, $order" | tee -a $OUT
    FGOICP_LIB=$DEV FGOICP_RCCL_LOAD_GLOBAL=$legacy LD_PRELOAD=$PWD/tools/abort_bt.so timeout -k 10 120 python3 tools/exit_probe.py $order >> $OUT 2>&1
    echo "exit status $?" | tee -a $OUT
  done
done
grep -v amdgpu.ids $OUT | grep -E "==|exit status|double free|rccl:|librocm|libamd" | head -40
