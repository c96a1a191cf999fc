set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/t7.log 2>&1; echo "pytest exit $?" >> gpurun_out/t7.log
tail -15 gpurun_out/t7.log
grep -q "pytest exit 0" gpurun_out/t7.log || exit 1
timeout -k 10 300 python tests/gpu_nn_timing.py bunny > gpurun_out/nn3.log 2>&1; cat gpurun_out/nn3.log
(timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench5.log 2>&1; echo "exit $?" >> gpurun_out/bench5.log)
tail -3 gpurun_out/bench5.log
