cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/t13.log 2>&1; echo "pytest exit $?" >> gpurun_out/t13.log
tail -40 gpurun_out/t13.log
