cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_edge_cases.py tests/test_gpu_rank_deficient.py tests/test_trimming.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r3_t7.txt 2>&1; echo "pytest rc $?"; tail -n 4 gpurun_out/r3_t7.txt
python tools/icp_bench.py bunny 5 2>&1 | grep '"default"' | cut -c40-175
FGOICP_ICP_GATED=0 python tools/icp_bench.py bunny 5 2>&1 | grep '"default"' | cut -c40-175 | sed 's/^/gated=0 /'
FGOICP_ICP_GATED=0 python bench.py --only default_threshold 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['reference_default_threshold']; print('gated=0 default_threshold ms', round(d['default_threshold_ms_per_step'],2), 'icp s', r['seconds_icp_rank0'])"
python bench.py --only default_threshold 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['reference_default_threshold']; print('default_threshold ms', round(d['default_threshold_ms_per_step'],2), 'icp s', r['seconds_icp_rank0'], d.get('icp_latency',{}).get('us_per_iteration'))"
python bench.py --only dragon 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['dragon_shape']; print('dragon wall', r['wall_clock_to_optimum_s'], 'icp s', r['seconds_icp_rank0'], r['icp_latency']['us_per_iteration'])"
python bench.py --only trimmed 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['trimmed_1m_outliers']; print('trimmed wall', r['wall_clock_to_optimum_s'], 'icp s', r['seconds_icp_rank0'], r['icp_latency']['us_per_iteration'])"
