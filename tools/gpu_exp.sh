cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t38.log 2>&1; echo "exit $?" >> gpurun_out/t38.log
tail -4 gpurun_out/t38.log
grep -q "exit 0" gpurun_out/t38.log || exit 1
B="python bench.py --no-cpu-baseline --no-dragon --no-trimmed --steps 3 --warmup 1"
for i in 1 2; do (timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"seconds_icp_rank0": [0-9.]*\|"wall_clock_to_optimum_s": [0-9.]*' | tr '\n' ' '); echo; done
(timeout -k 10 200 python tools/run_probe.py 0 2>&1 | grep -o '"seconds": [0-9.]*\|"seconds_icp": [0-9.]*' | tr '\n' ' '); echo " dragon"
