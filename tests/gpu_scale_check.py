"""Not a test: one full run per large configuration (robustness at scale, timings)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import fgoicp_amd as fg

def run(name, res, mse, K, trim=0.0, **kw):
    tgt, src, R_gt, t_gt = fg.synth.workload(name, angle_deg=150.0, min_angle_deg=110.0, **kw)
    t0 = time.perf_counter()
    s = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=K, trim_fraction=trim)
    t1 = time.perf_counter()
    R, t = s.run()
    t2 = time.perf_counter()
    st = s.stats()
    err = np.degrees(np.arccos(np.clip((np.trace(R.astype(np.float64).T @ R_gt) - 1) / 2, -1, 1)))
    print(f"[{name} nt={len(tgt)} ns={len(src)} res={res} mse={mse} K={K} trim={trim}] setup {t1-t0:.2f}s run {t2-t1:.3f}s lut={s.registration.lut_dims()} "
          f"sse={s.get_best_error():.4f} rot_err={err:.3f}deg t_err={np.linalg.norm(t-t_gt):.2e} subcubes={st['trans_cubes']} rot_cubes={st['rot_cubes']} "
          f"icp_runs={st['icp_runs']} icp_s={st['seconds_icp']:.3f} -> {st['trans_cubes']/(t2-t1):.0f} subcubes/s", flush=True)
    s.close()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "bunny002"):
    run("bunny", 0.002, 1e-3, 32)          # test/bunny.toml resolution: 2.4 GB LUT
if which in ("all", "dragon"):
    run("dragon", 0.005, 1e-3, 32)
    run("dragon", 0.005, 5e-6, 32)         # certify regime at 437k points (ns*mse = 2.2)
if which in ("all", "1m"):
    run("synthetic1m", 0.005, 1e-3, 32)
if which in ("all", "trim"):
    run("synthetic1m_outliers", 0.005, 1e-3, 32, trim=0.2)   # BASELINE config 5 on one GPU
    run("synthetic1m_outliers", 0.005, 1e-3, 32, trim=0.0)   # the same clouds without trimming (reference behaviour)
    run("bunny", 0.005, 5e-5, 32, trim=0.1)                  # certify regime with trimming: cost of the selection passes
