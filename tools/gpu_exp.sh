cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(REPLAY_ONLY_RANK=0 FGOICP_TIMING=1 timeout -k 10 500 python tools/scale_replay.py 4 bunny 5e-5 0.005 2>&1 | grep "timing\] round\|timing\] run\|estimated" | tail -8 | cut -c1-200 | sed 's/.*round \([0-9]*\):.*submissions \([0-9]*\),.*tasks \([0-9.]*\) ms.*round \([0-9.]*\) ms.*/r\1 sub \2 tasks \3 round \4/' )
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 3 --warmup 1"
for i in 1 2; do (FGOICP_TIMING=1 timeout -k 10 200 $B 2>&1 | grep "timing\] run\|\"value\"" | tail -2 | cut -c1-190 | tr '\n' ' '); echo; done
