cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1"
for S in 0 1 0 1; do (FGOICP_SORT_RANKS=$S timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_us": [0-9.]*\|"achieved": [0-9.]*' | tr '\n' ' '); echo " ranks=$S"; done
for S in 0 1; do (FGOICP_SORT_RANKS=$S timeout -k 10 200 python tools/run_probe.py 0 2>&1 | grep -o '"seconds": [0-9.]*\|"kernel_ms": [0-9.]*' | tr '\n' ' '); echo " dragon ranks=$S"; done
