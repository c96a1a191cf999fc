# Development build with -DFGOICP_SCAN_STATS: what the waves of the exact-NN scan spend their steps on (per walk: candidate top boxes,
# super-leaves, leaves tested against the per-query bound, leaves scanned).   bash tools/scan_stats.sh   (on the GPU box)
cd $GRAFT_REPO_ROOT
MODE=${1:-1}   # 1: counters, 2: cycle stamps only
LIB=/tmp/libfgoicp_stats$MODE.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip -DFGOICP_SCAN_STATS=$MODE -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -shared -o $LIB \
   fast-go-icp_amd/csrc/device/kernels.hip fast-go-icp_amd/csrc/device/ctx.hip fast-go-icp_amd/csrc/device/bvh.hip fast-go-icp_amd/csrc/host/solver.cpp fast-go-icp_amd/csrc/host/multi.cpp -ldl 2> gpurun_out/scan_stats_build.err || { tail -20 gpurun_out/scan_stats_build.err; exit 1; }
FGOICP_LIB=$LIB python - <<'PY'
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import fgoicp_amd as fg
lib = C.CDLL(os.environ["FGOICP_LIB"])
def stats(reset=True):
    a = (C.c_ulonglong * 8)(); lib.fgoicp_debug_scan_stats(a, int(reset)); return list(a)
for wl in ("bunny", "dragon"):
    tgt, src, R_gt, t_gt = fg.synth.workload(wl, angle_deg=150.0, min_angle_deg=110.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    reg = fg.Registration(pct, pcs, bounds, 0.005)
    stats()
    for label, R0, iters in (("far (first 6 iterations from identity, clouds 110-150 deg apart)", np.eye(3), 6),):
        icp = fg.IterativeClosestPoint3D(reg, None, None, iters, 0.0, R0, np.zeros(3)); icp.run()
        s = stats()
        w = max(s[0], 1)
        print(f"{wl} {label}: walks {s[0]}, per walk: top candidates {s[1]/w:.2f}, super-leaf candidates {s[2]/w:.2f}, leaves tested {s[3]/w:.2f}, leaves scanned {s[4]/w:.2f}; MAX over walks: leaves scanned {s[5]}, super-leaves {s[6]}; walks with > 16 leaves scanned: {s[7]}")
    sse, R, t = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.0005, np.eye(3), np.zeros(3)).run()
    stats()
    icp = fg.IterativeClosestPoint3D(reg, None, None, 4, 0.0, R, t); icp.run()
    s = stats(); w = max(s[0], 1)
    print(f"{wl} converged (4 iterations from the ICP's fixed point): walks {s[0]}, per walk: top candidates {s[1]/w:.2f}, super-leaf candidates {s[2]/w:.2f}, leaves tested {s[3]/w:.2f}, leaves scanned {s[4]/w:.2f}; MAX over walks: leaves scanned {s[5]}, super-leaves {s[6]}; walks with > 16 leaves scanned: {s[7]}")
    # where the time of ONE index-mode scan goes (s_memtime stamps of wave 0 of every block: entry, seeds loaded, first walk done, exit)
    nb = min(4096, (len(pcs) + 63) // 64)
    for label, (RR, tt) in (("far", (np.eye(3, dtype=np.float32), np.zeros(3, np.float32))), ("converged", (R, t))):
        w = (pcs @ np.asarray(RR, np.float32).T + np.asarray(tt, np.float32)).astype(np.float32)
        reg.procrustes(w); reg.procrustes(w)
        buf = (C.c_ulonglong * (4 * nb))(); lib.fgoicp_debug_scan_times(buf, nb)
        a = np.array(list(buf), dtype=np.float64).reshape(nb, 4)
        a = a[a[:, 0] > 0]
        t0 = a[:, 0].min()
        print(f"{wl} {label}, one index scan, {nb} blocks, shader cycles: first block starts at 0, last block starts at {a[:,0].max()-t0:.0f}, last exit {a[:,3].max()-t0:.0f}; "
              f"per block (mean / p99 / max): seeds {np.mean(a[:,1]-a[:,0]):.0f} / {np.percentile(a[:,1]-a[:,0],99):.0f} / {np.max(a[:,1]-a[:,0]):.0f}, "
              f"walk {np.mean(a[:,2]-a[:,1]):.0f} / {np.percentile(a[:,2]-a[:,1],99):.0f} / {np.max(a[:,2]-a[:,1]):.0f}, "
              f"combine+tail {np.mean(a[:,3]-a[:,2]):.0f} / {np.percentile(a[:,3]-a[:,2],99):.0f} / {np.max(a[:,3]-a[:,2]):.0f}")
    reg.close()
PY
