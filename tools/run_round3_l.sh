cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03l_gputests.log 2>&1; echo "exit $?" >> gpurun_out/r03l_gputests.log)
tail -3 gpurun_out/r03l_gputests.log
grep -q '^exit 0' gpurun_out/r03l_gputests.log || exit 1
leg() {
  python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='' else d['$2']
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes', int(r.get('subcubes_per_step', d.get('subcubes_per_step',0))), 'icp ms', round(r['seconds_icp_rank0']*1e3,2), 'best_sse', r.get('best_sse', (d.get('result') or {}).get('best_sse')))"
}
for i in 1 2; do leg default_threshold reference_default_threshold; leg headline ""; leg dragon dragon_shape; done
python tools/scale_replay.py 8 bunny 5e-5 0.005 2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  bunny W=8 x', round(d['estimated_speedup'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'subcubes', sum(d['subcubes_rank']))"
