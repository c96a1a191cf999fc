cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_cli_dist.py -m gpu -x -q -k "facades" 2>&1 | tail -25
