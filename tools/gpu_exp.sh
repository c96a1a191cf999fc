cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/ab2.log
FGOICP_LUT_ZPAIR=2 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1 >> gpurun_out/ab2.log
for z in 1 2; do echo "layout $z" >> gpurun_out/ab2.log; FGOICP_LUT_ZPAIR=$z timeout -k 10 120 python tests/gpu_microbench.py bunny 0.005 siblings 2>&1 | grep "G= 64" >> gpurun_out/ab2.log; FGOICP_LUT_ZPAIR=$z timeout -k 10 120 python tests/gpu_microbench.py dragon 0.005 random 2>&1 | grep "G= 64" >> gpurun_out/ab2.log; done
for rep in 1 2; do for z in 1 2; do
  echo "bench layout $z" >> gpurun_out/ab2.log
  FGOICP_LUT_ZPAIR=$z timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-default-threshold-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])" >> gpurun_out/ab2.log
done; done
cat gpurun_out/ab2.log
