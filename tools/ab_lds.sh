# A/B of the LDS-staged LUT tiles (bounds_lds_kernel) on the dense legs:  bash tools/ab_lds.sh
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r3_t5.txt 2>&1; echo "pytest rc $?"; tail -n 8 gpurun_out/r3_t5.txt
for L in 0 128 192; do
  for LEG in trimmed dragon; do
    FGOICP_LDS_STATS=1 FGOICP_LDS_TILES=$L timeout -k 10 300 python bench.py --only $LEG > gpurun_out/r3_lds_${LEG}_$L.log 2>&1
    grep "lds tiles" gpurun_out/r3_lds_${LEG}_$L.log | tail -n 1
    python - <<PY
import json
l=[x for x in open('gpurun_out/r3_lds_${LEG}_$L.log') if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); k='trimmed_1m_outliers' if '$LEG'=='trimmed' else 'dragon_shape'
    s=d[k]; r=s['roofline']
    print('lds_rows=$L', '$LEG', 'wall', round(s['wall_clock_to_optimum_s'],4), 'bnb', round(s['seconds_bnb_rank0'],4), 'subcubes', s['subcubes_per_step'], 'kernel_us', round(r['avg_launch_us'],1), 'launches', r['launches'], 'sse', s['best_sse'])
else:
    print('lds_rows=$L $LEG: no line')
PY
  done
done
