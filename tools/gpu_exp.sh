cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "dragon_size" 2>&1 | tail -2
(timeout -k 10 400 python bench.py > gpurun_out/bench11.log 2>&1; echo "exit $?" >> gpurun_out/bench11.log)
tail -2 gpurun_out/bench11.log | cut -c1-200
