cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 300 python bench.py --schedule serial --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"subcubes_per_step": [0-9.]*'| tr '\n' ' '); echo
