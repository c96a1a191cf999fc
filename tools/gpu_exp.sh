cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
bash tools/gpu_profile.sh r01
export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
rm -rf /tmp/ktrace
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-default-threshold-run --no-dragon > $REPO/gpurun_out/ktrace.log 2>&1
cd $REPO/tools && python trace_gaps.py /tmp/ktrace ../gpurun_out/profiles/r01_bench_trace_gaps.json > ../gpurun_out/trace_gaps.log 2>&1
