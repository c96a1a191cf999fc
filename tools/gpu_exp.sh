cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_cli_dist.py -m gpu -x -q > gpurun_out/t9.log 2>&1; echo "pytest exit $?" >> gpurun_out/t9.log
tail -25 gpurun_out/t9.log
timeout -k 10 700 python tests/gpu_scale_check.py all > gpurun_out/scale1.log 2>&1; echo "exit $?" >> gpurun_out/scale1.log
cat gpurun_out/scale1.log
