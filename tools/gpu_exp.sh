# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
(timeout -k 10 400 python bench.py > gpurun_out/bench_final.log 2>&1; echo "exit $?" >> gpurun_out/bench_final.log)
tail -2 gpurun_out/bench_final.log | cut -c1-260
