cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t27.log 2>&1; echo "exit $?" >> gpurun_out/t27.log
tail -4 gpurun_out/t27.log
grep -q "exit 0" gpurun_out/t27.log || exit 1
for M in 2 1; do
(FGOICP_SERIAL_SPECULATE=$M FGOICP_TIMING=1 timeout -k 10 300 python bench.py --schedule serial --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 2 --warmup 1 2>&1 | grep "timing\] run\|value" | cut -c1-330 | tail -2) >> gpurun_out/serial2.log 2>&1
done
cat gpurun_out/serial2.log
