cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
ARGS="$REPO/tools/run_probe.py 0"
cd /tmp
rm -rf /tmp/p?
i=0
pass() { i=$((i+1)); timeout -k 10 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/p$i -- python3 $ARGS > $REPO/gpurun_out/pmcd_$i.log 2>&1; }
pass TA_BUSY_avr GRBM_GUI_ACTIVE &&
pass TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
pass TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum &&
pass TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum &&
pass SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU &&
pass SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SALU
cd $REPO && python tools/pmc_generic.py gpurun_out/dragon_pmc.json /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5 /tmp/p6 > gpurun_out/dragon_pmc.txt 2>&1
cat gpurun_out/dragon_pmc.txt
