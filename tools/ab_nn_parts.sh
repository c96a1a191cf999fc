cd $GRAFT_REPO_ROOT
for P in 4 2 1 8; do
export FGOICP_NN_PARTS=$P
python bench.py --only trimmed 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['trimmed_1m_outliers']; print('parts=$P trimmed wall', round(r['wall_clock_to_optimum_s'],4), 'icp s', round(r['seconds_icp_rank0'],4))"
python bench.py --only dragon 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['dragon_shape']; print('parts=$P dragon wall', round(r['wall_clock_to_optimum_s'],4), 'icp s', round(r['seconds_icp_rank0'],4))"
done
