cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t23.log 2>&1; echo "exit $?" >> gpurun_out/t23.log
tail -4 gpurun_out/t23.log
grep -q "exit 0" gpurun_out/t23.log || exit 1
FGOICP_SMALL_TICK=0 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1"
run() { echo "== $*" >> gpurun_out/exp23.log; (env "$@" timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*' | tr '\n' ' ' >> gpurun_out/exp23.log); echo >> gpurun_out/exp23.log; }
rm -f gpurun_out/exp23.log
run FGOICP_SMALL_TICK=0
run FGOICP_SMALL_TICK=4096
run FGOICP_SMALL_TICK=32768
run FGOICP_SMALL_TICK=0
run FGOICP_SMALL_TICK=4096
cat gpurun_out/exp23.log
for W in 4; do timeout -k 10 300 python tools/dist_balance.py $W bunny 5e-5 0 2>&1 | tail -3; FGOICP_SMALL_TICK=0 timeout -k 10 300 python tools/dist_balance.py $W bunny 5e-5 0 2>&1 | tail -1; done
