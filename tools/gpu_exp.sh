cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t16.log 2>&1; echo "pytest exit $?" >> gpurun_out/t16.log
tail -4 gpurun_out/t16.log
rm -f gpurun_out/ab3.log
for spec in 1 0; do
  echo "serial speculate=$spec" >> gpurun_out/ab3.log
  FGOICP_SERIAL_SPECULATE=$spec timeout -k 10 300 python bench.py --schedule serial --steps 1 --warmup 1 --no-cpu-baseline --no-default-threshold-run 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['subcubes_per_step'], d['rot_cubes_rank0'], d['result'])" >> gpurun_out/ab3.log
done
cat gpurun_out/ab3.log
