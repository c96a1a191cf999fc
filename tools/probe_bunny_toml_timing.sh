# Probe: host-side timing of the bunny.toml-shape runs (SERIAL with / without its look-ahead, ROUND); FGOICP_TIMING lines on stderr.
cd $GRAFT_REPO_ROOT
for A in 480 0 480 0; do
echo "== FGOICP_SERIAL_AHEAD=$A"
FGOICP_SERIAL_AHEAD=$A FGOICP_TIMING=1 python - <<'PY' 2>&1 | grep -E 'timing\] (run|prepare)|wall' | tail -3
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fgoicp_amd as fg
tgt, src, _, _ = fg.synth.workload("bunny_toml", angle_deg=150.0, min_angle_deg=110.0)
s = fg.FastGoICP(tgt, src, 0.005, 1e-4, schedule=fg.SCHEDULE_SERIAL)
s.run()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); s.run(); best = min(best, time.perf_counter() - t0)
print("serial wall", round(best*1e3,1), "ms", file=sys.stderr)
s.close()
PY
done
