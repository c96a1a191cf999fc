# A/B (round 3): look-ahead of SERIAL tasks in the tail of an evaluation (FGOICP_SERIAL_AHEAD = most nodes a task evaluates ahead of its pops; 0 = off).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_serial_ahead.txt
: > $OUT
for A in 0 480 96 224 0 480; do
  echo "== FGOICP_SERIAL_AHEAD=$A" | tee -a $OUT
  FGOICP_SERIAL_AHEAD=$A python bench.py --only serial 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['serial_reference_order']; rf=r['roofline']
print('  serial: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes/s', round(r['subcubes_per_s']), 'subcubes', int(r['subcubes_per_step']), 'pops', r['rounds'], 'launches', rf['launches'], 'evaluations', round(rf['evaluations_per_launch']*rf['launches']), 'best_sse', r['best_sse'])" | tee -a $OUT
done
for A in 0 480; do
  for WL in "bunny 5e-5 0.005 2" "dragon 5e-6 0.005 1"; do
    echo "== FGOICP_SERIAL_AHEAD=$A, SERIAL on 8 ranks (replay), $WL" | tee -a $OUT
    FGOICP_SERIAL_AHEAD=$A FGOICP_REPLAY_SCHEDULE=serial python tools/scale_replay.py 8 $WL 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  x', round(d['estimated_speedup'],2), 'with coll', round(d['estimated_speedup_with_collectives'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'same', d['same_optimum'])" | tee -a $OUT
  done
done
