# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s --durations=20 -k "toml or tick_sort" > gpurun_out/r02b_tests.log 2>&1
tail -40 gpurun_out/r02b_tests.log
