#!/bin/bash
# Round 4, session D: what the dense (dragon) leg's bounds kernel is bound by, old kernel family vs item kernel (development build):
# kernel stats, SQ counters (instructions per launch), TCP counters; LAST, a single TA counter pass with a short timeout — if it is
# killed at its limit nothing else runs after it (the pass hung in round 3; its log is kept whatever happens).
set -u -o pipefail
cd "$GRAFT_REPO_ROOT"
REPO=$GRAFT_REPO_ROOT
mkdir -p $REPO/gpurun_out/profiles
export TMPDIR=/tmp
DEV=$REPO/fast-go-icp_amd/lib/libfgoicp_amd_dev.so
cd /tmp
SET1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
SET4="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"
for item in 0 1; do
  rm -rf /tmp/st_$item /tmp/p1_$item /tmp/p4_$item
  FGOICP_LIB=$DEV FGOICP_BOUNDS_ITEM=$item timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$item -- python3 $REPO/bench.py --only dragon --steps 1 --warmup 0 > $REPO/gpurun_out/r04d_stats_item$item.log 2>&1 || { echo "stats pass item=$item failed"; tail -3 $REPO/gpurun_out/r04d_stats_item$item.log; exit 1; }
  cp /tmp/st_$item/*/*kernel_stats.csv $REPO/gpurun_out/profiles/r04_dragon_item${item}_kernel_stats.csv
  head -4 $REPO/gpurun_out/profiles/r04_dragon_item${item}_kernel_stats.csv | cut -c1-200
  FGOICP_LIB=$DEV FGOICP_BOUNDS_ITEM=$item timeout -k 10 300 rocprofv3 --pmc $SET1 --kernel-trace --output-format csv -d /tmp/p1_$item -- python3 $REPO/bench.py --only dragon --steps 1 --warmup 0 > $REPO/gpurun_out/r04d_px1_item$item.log 2>&1 || { echo "SQ pass item=$item failed"; exit 1; }
  FGOICP_LIB=$DEV FGOICP_BOUNDS_ITEM=$item timeout -k 10 300 rocprofv3 --pmc $SET4 --kernel-trace --output-format csv -d /tmp/p4_$item -- python3 $REPO/bench.py --only dragon --steps 1 --warmup 0 > $REPO/gpurun_out/r04d_px4_item$item.log 2>&1 || { echo "TCP pass item=$item failed"; exit 1; }
  python3 $REPO/tools/pmc_generic.py $REPO/gpurun_out/profiles/r04_dragon_item${item}_pmc_sq_tcp.json /tmp/p1_$item /tmp/p4_$item > $REPO/gpurun_out/r04d_pmc_item$item.txt 2>&1
  grep -E "^bounds_(sorted|item)_kernel" $REPO/gpurun_out/r04d_pmc_item$item.txt
done
# the TA pass, once, one counter, the shipped library, the smallest leg first
for spec in "default_threshold TA_BUSY_avr" "dragon TA_BUSY_avr" "dragon TA_ADDR_STALLED_BY_TC_CYCLES_sum" "dragon TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  set -- $spec
  rm -rf /tmp/ta_$1_$2
  echo "== TA pass: leg $1 counter $2"
  timeout -k 10 150 rocprofv3 --pmc $2 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/ta_$1_$2 -- python3 $REPO/bench.py --only $1 --steps 1 --warmup 0 > $REPO/gpurun_out/r04d_ta_$1_$2.log 2>&1
  rc=$?
  echo "exit status $rc" >> $REPO/gpurun_out/r04d_ta_$1_$2.log
  if [ $rc -ne 0 ]; then echo "TA pass ($spec) ended with status $rc: the attempt ends here, log kept in gpurun_out/r04d_ta_$1_$2.log"; tail -5 $REPO/gpurun_out/r04d_ta_$1_$2.log | cut -c1-300; exit 0; fi
  python3 $REPO/tools/pmc_generic.py $REPO/gpurun_out/profiles/r04_$1_$2.json /tmp/ta_$1_$2 | grep -E "bounds_(sorted|item)_kernel"
done
