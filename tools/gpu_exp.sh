cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/scale5.log
timeout -k 10 600 python tools/scale_replay.py 5 bunny 5e-5 0.005 >> gpurun_out/scale5.log 2>&1 || exit 1
timeout -k 10 900 python tools/scale_replay.py 4 dragon 5e-6 0.005 >> gpurun_out/scale5.log 2>&1 || exit 1
cat gpurun_out/scale5.log | cut -c1-700
