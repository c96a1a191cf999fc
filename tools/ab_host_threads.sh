# A/B (round 3, final tree): worker threads of the host driver (FGOICP_HOST_THREADS; default 4) on the GPU box's CPU share.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/r03_ab_host_threads.txt
: > $OUT
nproc | tee -a $OUT
for T in 4 8 12 4 8; do
  echo "== FGOICP_HOST_THREADS=$T" | tee -a $OUT
  for LEG in "headline -" "serial serial_reference_order" "default_threshold reference_default_threshold"; do
    set -- $LEG
    FGOICP_HOST_THREADS=$T python bench.py --only $1 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=d if '$2'=='-' else d['$2']
print('  $1: wall', round(r['wall_clock_to_optimum_s']*1e3,2), 'ms')" | tee -a $OUT
  done
  FGOICP_HOST_THREADS=$T python - <<'PY' 2>&1 | tee -a $OUT
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fgoicp_amd as fg
tgt, src, _, _ = fg.synth.workload("bunny_toml", angle_deg=150.0, min_angle_deg=110.0)
for sched, nm in ((fg.SCHEDULE_SERIAL, "serial"), (fg.SCHEDULE_ROUND, "round")):
    s = fg.FastGoICP(tgt, src, 0.005, 1e-4, schedule=sched, round_width=0 if sched == fg.SCHEDULE_ROUND else 1)
    s.run(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); s.run(); best = min(best, time.perf_counter() - t0)
    print(f"  bunny_toml shape {nm}: wall {best*1e3:.1f} ms")
    s.close()
PY
done
