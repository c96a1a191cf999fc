# scratch script for ad-hoc GPU experiments (edited per experiment; see tools/gpu_profile.sh for the round profile)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for U in 0 1; do
echo "upload_kernel $U toml: $(PROBE_RES=0.002 FGOICP_UPLOAD_KERNEL=$U FGOICP_TIMING=1 timeout -k 10 200 python tools/run_probe.py 0 1e-4 bunny_toml 2>&1 | grep -E "timing\] (run|ticks)" | cut -c1-150 | tr '\n' ' ')"
done; done
for U in 0 1; do
echo "upload_kernel $U bunny: $(FGOICP_UPLOAD_KERNEL=$U FGOICP_TIMING=1 timeout -k 10 200 python tools/run_probe.py 0 5e-5 bunny 2>&1 | grep -E "timing\] (run|ticks)" | cut -c1-150 | tr '\n' ' ')"
done
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
