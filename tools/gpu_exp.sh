cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t30.log 2>&1; echo "exit $?" >> gpurun_out/t30.log
tail -4 gpurun_out/t30.log
grep -q "exit 0" gpurun_out/t30.log || exit 1
bash tools/gpu_profile.sh r01 > gpurun_out/profile_run.log 2>&1
tail -3 gpurun_out/profile_run.log
grep '^{"metric"' gpurun_out/bench_default.log | cut -c1-250
