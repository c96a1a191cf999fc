"""One run on a synthetic pair: python tools/run_probe.py <round_width> [mse] [workload] [trim_fraction] [angle_deg]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fgoicp_amd as fg
K = int(sys.argv[1]); mse = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-6
wl = sys.argv[3] if len(sys.argv) > 3 else "dragon"
trim = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
ang = float(sys.argv[5]) if len(sys.argv) > 5 else 150.0
tgt, src, R_gt, t_gt = fg.synth.workload(wl, angle_deg=ang, min_angle_deg=min(110.0, ang * 0.7))
t0 = time.perf_counter()
res = float(os.environ.get('PROBE_RES', '0.005'))
s = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=K, device=0, trim_fraction=trim)
setup = time.perf_counter() - t0
reg = s.registration; reg.set_profile(True); reg.profile(reset=True)
t0 = time.perf_counter(); R, t = s.run(); e = time.perf_counter() - t0
p = reg.profile(reset=True); st = s.stats()
err = float(np.degrees(np.arccos(np.clip((np.trace(R.astype(np.float64).T @ R_gt) - 1) / 2, -1, 1))))
print(json.dumps({"workload": wl, "trim": trim, "setup_s": setup, "rot_err_deg": err, "K": K, "seconds": e, "subcubes": st["trans_cubes"], "subcubes_per_s": st["trans_cubes"] / e, "rounds": st["rounds"], "rot_cubes": st["rot_cubes"],
                  "bounds_calls": st["bounds_calls"], "icp_runs": st["icp_runs"], "icp_iters": st["icp_iters"], "seconds_icp": st["seconds_icp"], "kernel_ms": p["kernel_ms"], "launches": p["launches"],
                  "best_sse": float(s.get_best_error())}), flush=True)
s.close()
