cd $GRAFT_REPO_ROOT
for CL in 1 0 1 0; do
export FGOICP_NN_CLAIM=$CL
echo "claim=$CL"
python tools/icp_bench.py bunny 5 2>&1 | grep default_fused | cut -c60-170
python bench.py --only default_threshold 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['reference_default_threshold']; print('default_threshold ms', round(d['default_threshold_ms_per_step'],2), 'icp s', r['seconds_icp_rank0'])"
python bench.py --only dragon 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['dragon_shape']; print('dragon wall', r['wall_clock_to_optimum_s'], 'icp s', r['seconds_icp_rank0'])"
python bench.py --only trimmed 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['trimmed_1m_outliers']; print('trimmed wall', r['wall_clock_to_optimum_s'], 'icp s', r['seconds_icp_rank0'])"
done
