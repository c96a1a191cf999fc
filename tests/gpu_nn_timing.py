"""Not a test: wall time of the exact-NN operators at converged / far alignments (tuning aid)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import fgoicp_amd as fg

name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
tgt, src, R_gt, t_gt = fg.synth.workload(name, angle_deg=20.0)
pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
for flags, label in [(0, "scan"), (fg.FLAG_BRUTE_FORCE_NN, "brute")]:
    reg = fg.Registration(pct, pcs, bounds, 0.005, flags=flags)
    icp = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.0005, np.eye(3), np.zeros(3))
    t0 = time.perf_counter(); sse, R, t = icp.run(); dt = time.perf_counter() - t0
    print(f"[{name} {label}] icp {icp.iterations} iters {dt*1e3:.1f} ms -> {dt/icp.iterations*1e6:.0f} us/iter, sse {sse}")
    for lab, (RR, tt) in {"converged": (R, t), "identity": (np.eye(3, dtype=np.float32), np.zeros(3, np.float32)),
                          "far": (np.eye(3, dtype=np.float32), np.array([1.5, 0, 0], np.float32))}.items():
        reg.compute_sse_error(RR, tt)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); reg.compute_sse_error(RR, tt); ts.append(time.perf_counter() - t0)
        w = (pcs @ RR.T + tt).astype(np.float32)
        reg.procrustes(w)
        tp = []
        for _ in range(5):
            t0 = time.perf_counter(); reg.procrustes(w); tp.append(time.perf_counter() - t0)
        print(f"    {lab:10s}: sse call min {min(ts)*1e6:8.1f} us   procrustes call (incl. upload) min {min(tp)*1e6:8.1f} us")
    reg.close()
