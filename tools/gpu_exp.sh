cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/scale6.log
for F in 1 0; do
echo "== finalize_side $F" >> gpurun_out/scale6.log
(FGOICP_FINALIZE_SIDE=$F REPLAY_ONLY_RANK=0 FGOICP_TIMING=1 timeout -k 10 500 python tools/scale_replay.py 4 bunny 5e-5 0.005 2>&1 | grep "timing\] round\|estimated" | tail -7 | cut -c1-200 | sed 's/.*round \([0-9]*\):.*submissions \([0-9]*\),.*tasks \([0-9.]*\) ms.*round \([0-9.]*\) ms.*/r\1 sub \2 tasks \3 round \4/' ) >> gpurun_out/scale6.log 2>&1
done
cat gpurun_out/scale6.log | cut -c1-300
