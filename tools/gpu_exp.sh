set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 500 python bench.py --steps 1 --warmup 0 > gpurun_out/bench1.log 2>&1; echo "exit $?" >> gpurun_out/bench1.log) 
tail -5 gpurun_out/bench1.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1; echo "pytest exit $?" >> gpurun_out/t3.log
tail -15 gpurun_out/t3.log
