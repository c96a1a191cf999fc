cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1"
for i in 1 2 3; do (FGOICP_TIMING=1 timeout -k 10 200 $B 2>&1 | grep "timing\] run\|\"value\"" | tail -2 | cut -c1-200 | tr '\n' ' '); echo; done
