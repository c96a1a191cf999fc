cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(FGOICP_TIMING=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 1 --warmup 1 2>&1 | grep "timing\] round\|timing\] run" | tail -9 | cut -c17-260) > gpurun_out/rounds.log 2>&1
cat gpurun_out/rounds.log
