cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1"
run() { echo "== $*" >> gpurun_out/exp22.log; (env "$@" timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*\|"avg_launch_us": [0-9.]*\|"launches": [0-9]*' | tr '\n' ' ' >> gpurun_out/exp22.log); echo >> gpurun_out/exp22.log; }
rm -f gpurun_out/exp22.log
run FGOICP_MAX_SUBCUBES=32768
run FGOICP_MAX_SUBCUBES=65536
run FGOICP_MAX_SUBCUBES=16384
run FGOICP_MAX_SUBCUBES=32768
cat gpurun_out/exp22.log
