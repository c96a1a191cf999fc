# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
for T in 1 2 4 8 16; do
echo "threads $T toml: $(PROBE_RES=0.002 FGOICP_HOST_THREADS=$T FGOICP_TIMING=1 timeout -k 10 200 python tools/run_probe.py 0 1e-4 bunny_toml 2>&1 | grep -E "timing\] (run|prepare)" | cut -c1-150 | tr '\n' ' ')"
done
for T in 4 8; do
echo "threads $T bunny: $(FGOICP_HOST_THREADS=$T FGOICP_TIMING=1 timeout -k 10 200 python tools/run_probe.py 0 5e-5 bunny 2>&1 | grep -E "timing\] run" | cut -c1-200)"
done
for W in 8; do FGOICP_HOST_THREADS=4 timeout -k 10 300 python tools/scale_replay.py $W bunny 5e-5 0.005 2 2>/dev/null | cut -c1-400; done
