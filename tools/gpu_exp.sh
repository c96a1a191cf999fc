cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for W in 64 256 1024 4096; do
echo "width $W" >> gpurun_out/serial3.log
(FGOICP_SERIAL_WIDTH=$W timeout -k 10 300 python bench.py --schedule serial --no-cpu-baseline --no-default-threshold-run --no-dragon --no-trimmed --steps 2 --warmup 1 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"subcubes_per_step": [0-9.]*' | tr '\n' ' ') >> gpurun_out/serial3.log 2>&1; echo >> gpurun_out/serial3.log
done
cat gpurun_out/serial3.log
