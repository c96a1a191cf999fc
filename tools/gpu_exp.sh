cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1"
run() { echo "== $*" >> gpurun_out/exp25.log; (env "$@" timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' >> gpurun_out/exp25.log); echo >> gpurun_out/exp25.log; }
rm -f gpurun_out/exp25.log
run A=1
run BENCH_NO_PROFILE=1
run A=1
run BENCH_NO_PROFILE=1
cat gpurun_out/exp25.log
