cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --steps 3 --warmup 1"
(timeout -k 10 300 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"wall_clock_to_optimum_s": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' '); echo
