set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1; echo "pytest exit $?" >> gpurun_out/t6.log
tail -15 gpurun_out/t6.log
grep -q "pytest exit 0" gpurun_out/t6.log || exit 1
rm -f gpurun_out/mb5.log
for mode in random siblings; do timeout -k 10 120 python tests/gpu_microbench.py bunny 0.005 $mode >> gpurun_out/mb5.log 2>&1; done
FGOICP_BOUNDS_SORTED=0 timeout -k 10 120 python tests/gpu_microbench.py bunny 0.005 siblings >> gpurun_out/mb5.log 2>&1
timeout -k 10 200 python tests/gpu_microbench.py dragon 0.005 random >> gpurun_out/mb5.log 2>&1
cat gpurun_out/mb5.log
(timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench4.log 2>&1; echo "exit $?" >> gpurun_out/bench4.log)
tail -3 gpurun_out/bench4.log
