# A/B of the trimmed leg's per-row selection on the GPU box: two passes (FGOICP_TRIM_SAMPLE=0) against one pass steered by the
# row's sample, for several sample strides and bracket margins.  usage: gpurun -- bash tools/trim_probe.sh > gpurun_out/...
cd $GRAFT_REPO_ROOT
run() {
python bench.py --only trimmed 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['trimmed_1m_outliers']; ro=r['roofline']
print('$1 wall', round(r['wall_clock_to_optimum_s'],4), 'bnb-icp', round(r['seconds_bnb_rank0']-r['seconds_icp_rank0'],4), 'icp', round(r['seconds_icp_rank0'],4), 'subcubes', r['subcubes_per_step'],
      'bnb_rate', round(r['bnb_without_icp_subcubes_per_s_rank0']), 'bounds_us', round(ro['avg_launch_us'],1), 'select_ms', round(ro.get('select_kernel_ms',0),1), 'bounds_ms', round(ro.get('bounds_kernel_ms',0),1),
      'rows', ro.get('select_rows'), 'fallbacks', ro.get('select_rows_done_again_in_two_passes'), 'members/row', ro.get('select_bracket_members_per_row'), 'sse', r['best_sse'])"
}
FGOICP_TRIM_SAMPLE=0 run "two-pass      "
FGOICP_TRIM_SAMPLE=5 FGOICP_TRIM_MARGIN=3 run "1/32 margin 3 "
FGOICP_TRIM_SAMPLE=5 FGOICP_TRIM_MARGIN=2 run "1/32 margin 2 "
FGOICP_TRIM_SAMPLE=5 FGOICP_TRIM_MARGIN=1 run "1/32 margin 1 "
FGOICP_TRIM_SAMPLE=4 FGOICP_TRIM_MARGIN=3 run "1/16 margin 3 "
FGOICP_TRIM_SAMPLE=6 FGOICP_TRIM_MARGIN=3 run "1/64 margin 3 "
