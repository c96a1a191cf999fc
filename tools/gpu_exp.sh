cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t20.log 2>&1; echo "exit $?" >> gpurun_out/t20.log
tail -4 gpurun_out/t20.log
grep -q "exit 0" gpurun_out/t20.log || exit 1
(timeout -k 10 500 python bench.py > gpurun_out/bench20.log 2>&1; echo "exit $?" >> gpurun_out/bench20.log)
tail -2 gpurun_out/bench20.log | cut -c1-300
