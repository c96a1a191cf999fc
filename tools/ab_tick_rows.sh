# A/B (round 3): ROUND batches sized by live tasks so that a tick carries about FGOICP_TICK_ROWS rows (0 = the stepwise tail rule).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_tick_rows.txt
: > $OUT
for R in 0 4096 8192 16384 32768 0; do
  echo "== FGOICP_TICK_ROWS=$R" | tee -a $OUT
  for LEG in "default_threshold reference_default_threshold" "headline -"; do
    set -- $LEG
    FGOICP_TICK_ROWS=$R python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='-' else d['$2']
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes', int(r.get('subcubes_per_step', d.get('subcubes_per_step',0))), 'best_sse', r.get('best_sse', (d.get('result') or {}).get('best_sse')))" | tee -a $OUT
  done
  FGOICP_TICK_ROWS=$R python tools/scale_replay.py 8 bunny 5e-5 0.005 2 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  8-rank replay: x', round(d['estimated_speedup'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'mean', round(sum(d['T_rank_s'])/8*1e3,1), 'subcubes', sum(d['subcubes_rank']), 'same', d['same_optimum'])" | tee -a $OUT
done
