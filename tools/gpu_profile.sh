# Round profile: bench + rocprofv3 kernel stats + PMC passes of the SAME command, leg by leg (bench.py --only LEG).
# Raw traces stay in /tmp on the box; only summaries come back through gpurun_out/profiles/ (copy them into profiles/ afterwards).
#   bash tools/gpu_profile.sh <tag> <read_factor> [skip_bench]
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
FACTOR=${2:-2}
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/profiles
mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$3" ]; then
  (timeout -k 10 700 python bench.py > gpurun_out/${TAG}_bench_default.log 2>&1; echo "exit $?" >> gpurun_out/${TAG}_bench_default.log)
  grep '^{"metric"' gpurun_out/${TAG}_bench_default.log > $OUT/${TAG}_bench_line.json
  tail -2 gpurun_out/${TAG}_bench_default.log | cut -c1-300
fi
cd /tmp
SPECS=""
for LEG in headline dragon trimmed default_threshold; do
  rm -rf /tmp/st_$LEG /tmp/pf_$LEG /tmp/pw_$LEG
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$LEG -- python3 $REPO/bench.py --only $LEG --steps 3 --warmup 1 > $REPO/gpurun_out/${TAG}_stats_$LEG.log 2>&1 || exit 1
  cp /tmp/st_$LEG/*/*kernel_stats.csv $OUT/${TAG}_${LEG}_kernel_stats.csv
  grep '^{"metric"' $REPO/gpurun_out/${TAG}_stats_$LEG.log > $OUT/${TAG}_${LEG}_under_rocprof.json
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$LEG -- python3 $REPO/bench.py --only $LEG --steps 1 --warmup 1 > $REPO/gpurun_out/${TAG}_pmcf_$LEG.log 2>&1 || exit 1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/pw_$LEG -- python3 $REPO/bench.py --only $LEG --steps 1 --warmup 1 > $REPO/gpurun_out/${TAG}_pmcw_$LEG.log 2>&1 || exit 1
  SPECS="$SPECS $LEG=/tmp/pf_$LEG,/tmp/pw_$LEG"
  echo "leg $LEG profiled"
done
cd $REPO && python tools/pmc_summary.py $TAG $OUT $FACTOR $SPECS
ls -la $OUT
