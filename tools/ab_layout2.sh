# Packed-LUT layout of the sparse (bunny-shape) context: yz-quad runs (2, default) vs the apron-bricked quads (4) vs 2x2x2 bricks (3).
cd $GRAFT_REPO_ROOT
for Z in 2 4 3 2 4; do
  FGOICP_LUT_ZPAIR=$Z timeout -k 10 300 python bench.py --only headline > gpurun_out/r3b_layout_$Z.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r3b_layout_$Z.log') if x.startswith('{"metric"')]
d=json.loads(l[-1]); r=d['roofline']
print('layout=$Z', 'subcubes/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'kernel_us', round(r['avg_launch_us'],1), 'launches', r['launches'], 'algorithmic_GBps', round(r['achieved']), 'sse', d['result']['best_sse'])
PY
done
