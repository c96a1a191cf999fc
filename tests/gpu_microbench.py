"""Not a test: quick GPU timing of the operators on the bench workload (used while tuning)."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import fgoicp_amd as fg

name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
res = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
mode = sys.argv[3] if len(sys.argv) > 3 else "random"
tgt, src, R_gt, t_gt = fg.synth.workload(name, angle_deg=120.0, min_angle_deg=90.0)
t_c, s_c, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
t0 = time.time()
reg = fg.Registration(t_c, s_c, bounds, res, flags=fg.FLAG_PROFILE)
print(f"[{name} res={res} P={os.environ.get('FGOICP_PTS_PER_THREAD','auto')} mode={mode}] ctx_create (upload + LUT {reg.lut_dims()}): {time.time()-t0:.3f}s", flush=True)
rng = np.random.default_rng(0)
rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)

def batch(B):
    if mode == "siblings":  # 4 sibling groups of 8 children (span 0.0625) as the inner BnB pops them
        out = []
        for _ in range(B // 8):
            c = rng.uniform(-0.4, 0.4, 3)
            for j in range(8):
                out.append([c[0] - 0.0625 + (j & 1) * 0.125, c[1] - 0.0625 + (j >> 1 & 1) * 0.125, c[2] - 0.0625 + (j >> 2 & 1) * 0.125, 0.0625])
        return np.array(out, np.float32)
    return np.concatenate([rng.uniform(-0.5, 0.5, (B, 3)), np.full((B, 1), 0.125)], 1).astype(np.float32)

for G, B in [(1, 32), (16, 32), (64, 32)]:
    groups = [batch(B) for _ in range(G)]
    Rs = [rn.q.R] * G
    for _ in range(3):
        reg.compute_bounds_multi(Rs, [rn.span] * G, [False] * G, groups)
    reg.profile(reset=True)
    n = 20
    t0 = time.time()
    for _ in range(n):
        reg.compute_bounds_multi(Rs, [rn.span] * G, [False] * G, groups)
    dt = (time.time() - t0) / n
    p = reg.profile(reset=True)
    sub = G * B
    kern = p["kernel_ms"] / n * 1e-3  # bounds-kernel time per call (all launches of the call)
    bytes_sub = reg.ns * (32 + 12 / 32)
    print(f"G={G:3d} B={B}: wall {dt*1e6:8.1f} us/call  {sub/dt:12.0f} subcubes/s | kernel {kern*1e6:7.1f} us/call ({p['launches']//n} launches) "
          f"-> {sub/kern:12.0f} subcubes/s, {sub*bytes_sub/kern/1e9:8.1f} GB/s algorithmic", flush=True)
if "--ops" in sys.argv:
    for _ in range(2):
        t0 = time.time(); s = reg.compute_sse_error(np.eye(3), np.zeros(3)); dt = time.time() - t0
    print(f"exact sse: {dt*1e3:.2f} ms  (sse={s})")
    t0 = time.time(); icp = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.005, np.eye(3), np.zeros(3)); r = icp.run(); dt = time.time() - t0
    print(f"icp: {dt*1e3:.1f} ms, {icp.iterations} iters, sse={r[0]}")
