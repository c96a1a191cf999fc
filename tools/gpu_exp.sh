# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t37.log 2>&1; echo "exit $?" >> gpurun_out/t37.log
tail -3 gpurun_out/t37.log
grep -q "exit 0" gpurun_out/t37.log || exit 1
bash tools/gpu_profile.sh r01 > gpurun_out/profile_run.log 2>&1
grep '^{"metric"' gpurun_out/bench_default.log | cut -c1-200
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
ARGS="$REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed"
cd /tmp
rm -rf /tmp/p? /tmp/ktrace
i=0
pass() { i=$((i+1)); timeout -k 10 90 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/p$i -- python3 $ARGS > $REPO/gpurun_out/pmc_$i.log 2>&1; }
pass TA_BUSY_avr GRBM_GUI_ACTIVE &&
pass TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum &&
pass TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum &&
pass TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum &&
pass TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum &&
pass SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU &&
pass SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
cd $REPO && python tools/pmc_generic.py gpurun_out/profiles/r01_bounds_kernel_pmc_extra.json /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5 /tmp/p6 /tmp/p7 > gpurun_out/pmc_extra.txt 2>&1
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed > $REPO/gpurun_out/ktrace.log 2>&1
cd $REPO/tools && python trace_gaps.py /tmp/ktrace ../gpurun_out/profiles/r01_bench_trace_gaps.json 0.5 > ../gpurun_out/trace_gaps.log 2>&1
