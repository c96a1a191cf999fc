cd $GRAFT_REPO_ROOT
for V in "0 2" "1 2" "0 4" "1 4" "0 2" "1 2"; do
  set -- $V
  FGOICP_SORT_ORIENT=$1 FGOICP_LUT_ZPAIR=$2 timeout -k 10 300 python bench.py --only headline > gpurun_out/r3b_orient.log 2>&1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r3b_orient.log') if x.startswith('{"metric"')]
d=json.loads(l[-1]); r=d['roofline']
print('orient=$1 layout=$2', 'subcubes/s', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'kernel_us', round(r['avg_launch_us'],1), 'launches', r['launches'], 'algorithmic_GBps', round(r['achieved']), 'sse', d['result']['best_sse'])
PY
done
