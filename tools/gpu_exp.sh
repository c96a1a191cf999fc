cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
rm -f gpurun_out/mb6.log
timeout -k 10 120 python tests/gpu_microbench.py bunny 0.005 siblings >> gpurun_out/mb6.log 2>&1
timeout -k 10 200 python tests/gpu_microbench.py dragon 0.005 random >> gpurun_out/mb6.log 2>&1
cat gpurun_out/mb6.log
(timeout -k 10 400 python bench.py --steps 1 --warmup 0 --mse-threshold 5e-5 --round-width 4 --no-cpu-baseline > gpurun_out/bench_heavy4.log 2>&1; echo "exit $?" >> gpurun_out/bench_heavy4.log)
tail -2 gpurun_out/bench_heavy4.log | cut -c1-400
