cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time, subprocess, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
code = r'''
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fgoicp_amd as fg
name, mse = sys.argv[1], float(sys.argv[2])
tgt, src, _, _ = fg.synth.workload(name, angle_deg=150.0, min_angle_deg=110.0)
for sched, nm in ((fg.SCHEDULE_ROUND, "round"), (fg.SCHEDULE_SERIAL, "serial")):
    s = fg.FastGoICP(tgt, src, 0.005, mse, schedule=sched, round_width=0 if sched == fg.SCHEDULE_ROUND else 1)
    s.run()
    t0 = time.perf_counter(); s.run(); dt = time.perf_counter() - t0
    st = s.stats(); info = s.registration.info()
    print(f"  {name} {nm}: wall {dt*1e3:.1f} ms, subcubes {st['trans_cubes']}, {st['trans_cubes']/dt/1e6:.2f} M/s, icp {st['seconds_icp']*1e3:.1f} ms, sse {float(s.get_best_error()):.6f}, layout {info['lut_layout']}, pts/item {info['points_per_item']}, order {info['source_order']}")
    s.close()
'''
for name, mse in (("mid", "1.5e-5"), ("bunny_toml", "1e-4")):
    for env in ({"FGOICP_POINT_CURVE": "1", "FGOICP_BVH_ORDER": "0"}, {}):
        print("==", name, env or "defaults (k-d orders)")
        subprocess.run([sys.executable, "-c", code, name, mse], env={**os.environ, **env})
PY
