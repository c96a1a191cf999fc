set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/t5.log 2>&1; echo "pytest exit $?" >> gpurun_out/t5.log
tail -15 gpurun_out/t5.log
grep -q "pytest exit 0" gpurun_out/t5.log || exit 1
(timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench3.log 2>&1; echo "exit $?" >> gpurun_out/bench3.log)
tail -3 gpurun_out/bench3.log
timeout -k 10 300 python tests/gpu_microbench.py dragon 0.005 random --ops > gpurun_out/mb4.log 2>&1; cat gpurun_out/mb4.log
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r1 -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof_r1.log 2>&1
cd $REPO; find gpurun_out/prof_r1 -name "*stats*" | head; tail -2 gpurun_out/prof_r1.log
