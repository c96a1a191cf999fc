cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1"
for S in 1 2; do (timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_launch_us": [0-9.]*\|"achieved": [0-9.]*\|"best_sse": [0-9.]*' | tr '\n' ' '); echo; done
