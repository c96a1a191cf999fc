# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_cli_dist.py -m gpu -x -q --durations=8 2>&1 | tail -25
