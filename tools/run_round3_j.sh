cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 600 python -m pytest tests/test_trimming.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r03j_gputests.log 2>&1; echo "exit $?" >> gpurun_out/r03j_gputests.log)
tail -4 gpurun_out/r03j_gputests.log
grep -q '^exit 0' gpurun_out/r03j_gputests.log || exit 1
for M in 1 2; do
  echo "== trimmed leg, FGOICP_POINT_CURVE=$M (hashed sample positions)"
  FGOICP_POINT_CURVE=$M python bench.py --only trimmed 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['trimmed_1m_outliers']; rf=r['roofline']
print('  wall', round(r['wall_clock_to_optimum_s']*1e3,1), 'icp', round(r['seconds_icp_rank0']*1e3,1), 'bounds us', round(rf['avg_launch_us'],1), 'select ms', round(rf['select_kernel_ms'],1), 'rows', rf['select_rows'], 'fallbacks', rf['select_rows_done_again_in_two_passes'], 'members', round(rf['select_bracket_members_per_row'] or 0))"
done
timeout -k 10 900 bash tools/ab_waves.sh
