cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/cfg0.log
(PROBE_RES=0.002 FGOICP_TIMING=1 timeout -k 10 300 python tools/dragon_probe.py 0 1e-3 bunny_toml 0 150 >> gpurun_out/cfg0.log 2>&1; echo "exit $?" >> gpurun_out/cfg0.log)
(PROBE_RES=0.002 timeout -k 10 300 python tools/dragon_probe.py 1 1e-3 bunny_toml 0 150 >> gpurun_out/cfg0.log 2>&1; echo "exit $?" >> gpurun_out/cfg0.log)
grep -v amdgpu.ids gpurun_out/cfg0.log | cut -c1-700 | tail -12
