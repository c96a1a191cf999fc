# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_profile.sh r01 > gpurun_out/profile_run.log 2>&1
grep '^{"metric"' gpurun_out/bench_default.log | cut -c1-200
