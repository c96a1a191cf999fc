cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t39.log 2>&1; echo "exit $?" >> gpurun_out/t39.log
tail -4 gpurun_out/t39.log
grep -q "exit 0" gpurun_out/t39.log || exit 1
B="python bench.py --no-cpu-baseline --no-dragon --no-trimmed --steps 3 --warmup 1"
for L in 0 48 0 48 16 128; do (FGOICP_LOOKAHEAD=$L timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"wall_clock_to_optimum_s": [0-9.]*\|"launches": [0-9]*' | tr '\n' ' '); echo " lookahead=$L"; done
for L in 0 48; do
echo "== replay W=4 lookahead $L"
(FGOICP_LOOKAHEAD=$L REPLAY_ONLY_RANK=0 FGOICP_TIMING=1 timeout -k 10 500 python tools/scale_replay.py 4 bunny 5e-5 0.005 2>&1 | grep "timing\] round\|estimated" | tail -7 | cut -c1-200 | sed 's/.*round \([0-9]*\):.*submissions \([0-9]*\),.*tasks \([0-9.]*\) ms.*round \([0-9.]*\) ms.*/r\1 sub \2 tasks \3 round \4/' )
done
