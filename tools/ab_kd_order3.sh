# A/B (round 3), third pass: the density-split ("mixed") source order against the curve and the plain k-d order on the trimmed leg,
# and against the plain k-d order where there are no outliers (it must fall back to it).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_kd_order3.txt
: > $OUT
leg() {  # leg name, json key
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
rf=r.get('roofline') or {}
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, icp', round(r['seconds_icp_rank0']*1e3,2), 'ms, subcubes/s', round(r.get('subcubes_per_s', d['value'])), ', bounds kernel us', round(rf.get('avg_launch_us',0),1), 'setup s', round(r['setup_s_upload_plus_lut_build'],3))"
}
for M in 1 3 2 1 3; do export FGOICP_POINT_CURVE=$M; echo "== POINT_CURVE=$M" | tee -a $OUT; leg trimmed trimmed_1m_outliers | tee -a $OUT; done
export FGOICP_POINT_CURVE=3; echo "== POINT_CURVE=3, no outliers" | tee -a $OUT; leg headline "" | tee -a $OUT; leg dragon dragon_shape | tee -a $OUT
