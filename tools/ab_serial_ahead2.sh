# A/B (round 3): up to how many live tasks of a half the SERIAL look-ahead is on (FGOICP_SERIAL_AHEAD_TASKS = T; 480 / 224 / 96 nodes from T/4 / T/2 / T tasks down).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_serial_ahead2.txt
: > $OUT
for T in 32 128 512 1024 2048 4096 16384 512 2048; do
  echo "== FGOICP_SERIAL_AHEAD_TASKS=$T" | tee -a $OUT
  FGOICP_SERIAL_AHEAD_TASKS=$T python bench.py --only serial 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['serial_reference_order']; rf=r['roofline']
print('  serial: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms, subcubes/s', round(r['subcubes_per_s']), 'launches', rf['launches'], 'evaluations', round(rf['evaluations_per_launch']*rf['launches']), 'best_sse', r['best_sse'])" | tee -a $OUT
done
for T in 512 2048; do
  for WL in "bunny 5e-5 0.005 2" "dragon 5e-6 0.005 1"; do
  echo "== FGOICP_SERIAL_AHEAD_TASKS=$T, SERIAL on 8 ranks (replay), $WL" | tee -a $OUT
  FGOICP_SERIAL_AHEAD_TASKS=$T FGOICP_REPLAY_SCHEDULE=serial python tools/scale_replay.py 8 $WL 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('  x', round(d['estimated_speedup'],2), 'T1', round(d['T1_s'],3), 'slowest', round(max(d['T_rank_s'])*1e3,1), 'same', d['same_optimum'])" | tee -a $OUT
  done
done
echo "== dragon shape, one GPU, SERIAL" | tee -a $OUT
for T in 32 512 2048; do
FGOICP_SERIAL_AHEAD_TASKS=$T python - <<'PY' | tee -a $OUT
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fgoicp_amd as fg
tgt, src, _, _ = fg.synth.workload("dragon", angle_deg=150.0, min_angle_deg=110.0)
s = fg.FastGoICP(tgt, src, 0.005, 5e-6, schedule=fg.SCHEDULE_SERIAL)
t0 = time.perf_counter(); s.run(); dt = time.perf_counter() - t0
st = s.stats(); print("  T =", os.environ["FGOICP_SERIAL_AHEAD_TASKS"], "wall", round(dt, 3), "s, subcubes", st["trans_cubes"], "sse", float(s.get_best_error()))
PY
done
