cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t14.log 2>&1; echo "pytest exit $?" >> gpurun_out/t14.log
tail -30 gpurun_out/t14.log
