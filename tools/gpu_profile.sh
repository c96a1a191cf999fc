# Round profile: bench + rocprofv3 kernel stats + PMC passes of the SAME command (default bench flags).
# Raw traces stay in /tmp on the box; only summaries come back through gpurun_out/.
set -x
cd $GRAFT_REPO_ROOT
TAG=${1:-r01}
mkdir -p gpurun_out/profiles
(timeout -k 10 500 python bench.py > gpurun_out/bench_default.log 2>&1; echo "exit $?" >> gpurun_out/bench_default.log)
tail -2 gpurun_out/bench_default.log | cut -c1-300
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
CMD="python3 bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed"
rm -rf /tmp/prof_stats /tmp/pmc_fetch /tmp/pmc_write
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $REPO/bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed > $REPO/gpurun_out/prof_stats.log 2>&1
cp /tmp/prof_stats/*/*kernel_stats.csv $REPO/gpurun_out/profiles/${TAG}_bench_kernel_stats.csv
grep '^{"metric"' $REPO/gpurun_out/prof_stats.log > $REPO/gpurun_out/profiles/${TAG}_bench_under_rocprof.json
cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed > $REPO/gpurun_out/pmc_fetch.log 2>&1
cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed > $REPO/gpurun_out/pmc_write.log 2>&1
cd $REPO && python tools/pmc_summary.py /tmp/pmc_fetch /tmp/pmc_write $TAG "rocprofv3 --pmc ... -- $CMD --steps 1 --warmup 1" gpurun_out/profiles
ls -la gpurun_out/profiles
