cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --no-cpu-baseline --no-default-threshold-run --no-dragon --steps 3 --warmup 1"
run() { echo "== $*" >> gpurun_out/exp18.log; (env "$@" timeout -k 10 200 $B 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"achieved": [0-9.]*\|"seconds_icp_rank0": [0-9.]*' | tr '\n' ' ' >> gpurun_out/exp18.log); echo >> gpurun_out/exp18.log; }
rm -f gpurun_out/exp18.log
run FGOICP_POINT_CURVE=0
run FGOICP_POINT_CURVE=1
run FGOICP_POINT_CURVE=0
run FGOICP_POINT_CURVE=1
for V in 0 1; do
echo "dragon point curve $V" >> gpurun_out/exp18.log
(FGOICP_POINT_CURVE=$V timeout -k 10 200 python tools/dragon_probe.py 0 2>&1 | grep -o '"seconds": [0-9.]*\|"kernel_ms": [0-9.]*\|"seconds_icp": [0-9.]*' | tr '\n' ' ' >> gpurun_out/exp18.log); echo >> gpurun_out/exp18.log
done
cat gpurun_out/exp18.log
