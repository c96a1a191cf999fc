# A/B (round 3): k-d ordered BVH leaves (FGOICP_BVH_ORDER) and k-d ordered source cloud (FGOICP_POINT_CURVE=2) against the Hilbert runs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_kd_order.txt
: > $OUT
leg() {  # leg name, json key
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
rf=r.get('roofline') or {}
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, icp', round(r['seconds_icp_rank0']*1e3,2), 'ms, subcubes/s', round(r.get('subcubes_per_s', d['value'])), ', bounds kernel us', round(rf.get('avg_launch_us',0),1), 'setup s', round(r['setup_s_upload_plus_lut_build'],3))"
}
for CFG in "0 1" "1 1" "1 2" "0 1" "1 1" "1 2"; do
  set -- $CFG
  export FGOICP_BVH_ORDER=$1 FGOICP_POINT_CURVE=$2
  echo "== BVH_ORDER=$1 POINT_CURVE=$2" | tee -a $OUT
  for W in bunny dragon; do
    ICP_VARIANT=default python tools/icp_bench.py $W 5 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  icp_bench', d['workload'], 'thr', d['thr'], 'iters', d['iters'], 'us/iter', round(d['us_per_iter'],1))" | tee -a $OUT
  done
  leg default_threshold reference_default_threshold | tee -a $OUT
  leg headline "" | tee -a $OUT
  leg dragon dragon_shape | tee -a $OUT
  leg trimmed trimmed_1m_outliers | tee -a $OUT
done
