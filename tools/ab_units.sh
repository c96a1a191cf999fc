cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_trimming.py tests/test_gpu_multi.py tests/test_abi.py -x -q -m gpu > gpurun_out/r3_t4.txt 2>&1; echo "pytest rc $?"; tail -n 8 gpurun_out/r3_t4.txt
for U in 0 8 4; do
  for LEG in trimmed dragon; do
    FGOICP_UNITS=$U timeout -k 10 300 python bench.py --only $LEG > gpurun_out/r3_units_${LEG}_$U.log 2>&1
    python - <<PY
import json
l=[x for x in open('gpurun_out/r3_units_${LEG}_$U.log') if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); k='trimmed_1m_outliers' if '$LEG'=='trimmed' else 'dragon_shape'
    s=d[k]; r=s['roofline']
    print('units=$U', '$LEG', 'wall', round(s['wall_clock_to_optimum_s'],4), 'bnb', round(s['seconds_bnb_rank0'],4), 'icp', round(s['seconds_icp_rank0'],4), 'subcubes', s['subcubes_per_step'], 'kernel_us', round(r['avg_launch_us'],1), 'launches', r['launches'], 'sse', s['best_sse'])
else:
    print('units=$U $LEG: no line'); 
PY
  done
done
