cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/scale7.log
for W in 2 4 5; do timeout -k 10 600 python tools/scale_replay.py $W bunny 5e-5 0.005 >> gpurun_out/scale7.log 2>&1 || exit 1; done
timeout -k 10 900 python tools/scale_replay.py 4 dragon 5e-6 0.005 >> gpurun_out/scale7.log 2>&1 || exit 1
cut -c1-420 gpurun_out/scale7.log
