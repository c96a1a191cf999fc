cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t12.log 2>&1; echo "pytest exit $?" >> gpurun_out/t12.log
tail -5 gpurun_out/t12.log
(timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench8.log 2>&1; echo "exit $?" >> gpurun_out/bench8.log)
(FGOICP_PIPELINE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-default-threshold-run > gpurun_out/bench8_sync.log 2>&1; echo "exit $?" >> gpurun_out/bench8_sync.log)
(FGOICP_TIMING=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-default-threshold-run 2>&1 | grep timing > gpurun_out/bench8_timing.log)
cat gpurun_out/bench8_timing.log
