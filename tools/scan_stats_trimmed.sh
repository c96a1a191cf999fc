# Development build with -DFGOICP_SCAN_STATS: what the waves of the exact-NN scan spend their steps on in the TRIMMED 1M run's ICP
# (far from / at the optimum).   bash tools/scan_stats_trimmed.sh   (on the GPU box)
cd $GRAFT_REPO_ROOT
LIB=/tmp/libfgoicp_stats1.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -DFGOICP_SCAN_STATS=1 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -shared -o $LIB \
   fast-go-icp_amd/csrc/device/kernels.hip fast-go-icp_amd/csrc/device/ctx.hip fast-go-icp_amd/csrc/device/bvh.hip fast-go-icp_amd/csrc/host/solver.cpp fast-go-icp_amd/csrc/host/multi.cpp -ldl 2>/dev/null || exit 1
FGOICP_LIB=$LIB python - <<'PY'
import ctypes as C, numpy as np, sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fgoicp_amd as fg
lib = C.CDLL(os.environ["FGOICP_LIB"])
def stats(reset=True):
    a = (C.c_ulonglong * 8)(); lib.fgoicp_debug_scan_stats(a, int(reset)); return list(a)
def show(label, s, dt, iters):
    w = max(s[0], 1)
    print(f"{label}: {iters} iterations, {dt*1e3/max(iters,1):.2f} ms per iteration; walks {s[0]}, per walk: top candidates {s[1]/w:.2f}, super-leaf candidates {s[2]/w:.2f}, leaves tested {s[3]/w:.2f}, leaves scanned {s[4]/w:.2f}; MAX over walks: leaves scanned {s[5]}, super-leaves {s[6]}; walks with > 16 leaves scanned: {s[7]}", flush=True)
for wl, trim, flags in (("synthetic1m_outliers", 0.2, fg.FLAG_CURVE_ORDER), ("synthetic1m", 0.0, 0)):
    tgt, src, R_gt, t_gt = fg.synth.workload(wl, angle_deg=150.0, min_angle_deg=110.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    reg = fg.Registration(pct, pcs, bounds, 0.005, flags=flags)
    if trim: reg.set_inliers(int(len(pcs) * (1 - trim)))
    stats()
    icp = fg.IterativeClosestPoint3D(reg, None, None, 6, 0.0, np.eye(3), np.zeros(3))
    t0 = time.perf_counter(); icp.run(); dt = time.perf_counter() - t0
    show(f"{wl} trim {trim} FAR (first 6 iterations from identity)", stats(), dt, icp.iterations)
    sse, R, t = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.0005, np.eye(3), np.zeros(3)).run()
    stats()
    icp = fg.IterativeClosestPoint3D(reg, None, None, 4, 0.0, R, t)
    t0 = time.perf_counter(); icp.run(); dt = time.perf_counter() - t0
    show(f"{wl} trim {trim} at the ICP's fixed point from identity (sse {float(sse):.3f})", stats(), dt, icp.iterations)
    # at the ground truth
    Rg = R_gt.astype(np.float32); tg = ((t_gt + off_t * 0 ) ).astype(np.float32)
    reg.close()
PY
