# A/B (round 3), second pass under the k-d defaults: in-leaf order, chunk sizes, LUT layout.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_kd_order2.txt
: > $OUT
leg() {  # leg name, json key
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
rf=r.get('roofline') or {}
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, icp', round(r['seconds_icp_rank0']*1e3,2), 'ms, subcubes/s', round(r.get('subcubes_per_s', d['value'])), ', bounds kernel us', round(rf.get('avg_launch_us',0),1), 'setup s', round(r['setup_s_upload_plus_lut_build'],3))"
}
cfg() { echo "== $*" | tee -a $OUT; }
cfg defaults; leg default_threshold reference_default_threshold | tee -a $OUT; leg headline "" | tee -a $OUT; leg dragon dragon_shape | tee -a $OUT; leg trimmed trimmed_1m_outliers | tee -a $OUT
export FGOICP_KD_FINE=1; cfg KD_FINE=1; leg headline "" | tee -a $OUT; leg dragon dragon_shape | tee -a $OUT; leg default_threshold reference_default_threshold | tee -a $OUT; unset FGOICP_KD_FINE
for C in 128 512; do export FGOICP_CHUNK_PTS=$C; cfg CHUNK_PTS=$C; leg headline "" | tee -a $OUT; unset FGOICP_CHUNK_PTS; done
for C in 1024 4096; do export FGOICP_CHUNK_PTS=$C; cfg CHUNK_PTS=$C; leg dragon dragon_shape | tee -a $OUT; unset FGOICP_CHUNK_PTS; done
export FGOICP_LUT_ZPAIR=2; cfg LUT_ZPAIR=2 "(plain yz-quads)"; leg headline "" | tee -a $OUT; unset FGOICP_LUT_ZPAIR
export FGOICP_LUT_ZPAIR=2; cfg LUT_ZPAIR=2 "(yz-quads on the dense shape)"; leg dragon dragon_shape | tee -a $OUT; unset FGOICP_LUT_ZPAIR
export FGOICP_POINT_CURVE=2; cfg "trimmed with the k-d source order (flag overridden)"; leg trimmed trimmed_1m_outliers | tee -a $OUT; unset FGOICP_POINT_CURVE
cfg defaults again; leg headline "" | tee -a $OUT; leg dragon dragon_shape | tee -a $OUT
