"""Not a test (CPU only): what the exact-NN scan of csrc/device/kernels.hip (box_walk) has to test and scan per 64-query group at the optimum,
for the two orders of the target (runs of the Hilbert curve / cells of a k-d tree, csrc/device/morton.hpp) and of the queries.
    python tools/kd_sim.py bunny dragon"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fgoicp_amd as fg
from scipy.spatial import cKDTree

def expand10(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v

def hilbert30(x, y, z):
    X = [x.astype(np.uint32) & 1023, y.astype(np.uint32) & 1023, z.astype(np.uint32) & 1023]
    Q = 512
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            m = (X[i] & Q) != 0
            X0 = X[0].copy()
            # if bit set: X[0] ^= P ; else swap low bits of X[0], X[i]
            t = (X0 ^ X[i]) & P
            X[0] = np.where(m, X0 ^ P, X0 ^ t)
            if i != 0:
                X[i] = np.where(m, X[i], X[i] ^ t)
        Q >>= 1
    X[1] ^= X[0]; X[2] ^= X[1]
    t = np.zeros_like(X[0]); Q = 512
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ np.uint32(Q - 1), t); Q >>= 1
    X = [a ^ t for a in X]
    return expand10(X[2]) | (expand10(X[1]) << 1) | (expand10(X[0]) << 2)

def curve_order(p):
    lo = p.min(0); ext = (p.max(0) - lo).max()
    c = np.clip((p - lo) / ext * 1023.0, 0, 1023).astype(np.uint32)
    return np.argsort(hilbert30(c[:, 0], c[:, 1], c[:, 2]), kind='stable')

def kd_order(p, leaf=32):
    n = len(p); nleaf = (n + leaf - 1) // leaf
    depth = 0
    while (1 << depth) < nleaf: depth += 1
    perm = np.arange(n)
    def rec(lo, hi, cap_leaves):  # points perm[lo:hi] go into cap_leaves leaves
        if cap_leaves == 1 or hi - lo <= leaf: return
        half = cap_leaves // 2 * leaf
        if hi - lo <= half:
            rec(lo, hi, cap_leaves // 2); return
        pts = p[perm[lo:hi]]
        ax = np.argmax(pts.max(0) - pts.min(0))
        k = half
        idx = np.argpartition(pts[:, ax], k - 1)
        perm[lo:hi] = perm[lo:hi][idx]
        rec(lo, lo + k, cap_leaves // 2); rec(lo + k, hi, cap_leaves // 2)
    sys.setrecursionlimit(10000)
    rec(0, n, 1 << depth)
    return perm

def leaf_boxes(p, perm, leaf=32):
    n = len(p); nleaf = (n + leaf - 1) // leaf
    pad = nleaf * leaf - n
    q = p[perm]
    if pad: q = np.concatenate([q, np.repeat(q[-1:], pad, 0)])
    q = q.reshape(nleaf, leaf, 3)
    return q.min(1), q.max(1)

def box_d2(lo, hi, q):  # lo,hi (L,3), q (Q,3) -> (Q,L)
    d = np.maximum(np.maximum(lo[None] - q[:, None], q[:, None] - hi[None]), 0)
    return (d * d).sum(-1)

def simulate(name, ngroups=300, balanced=False):
    tgt, src, R_gt, t_gt = fg.synth.workload(name, angle_deg=150.0, min_angle_deg=110.0)
    tgt = tgt.astype(np.float64); q = src.astype(np.float64) @ R_gt.T + t_gt
    so = curve_order(src.astype(np.float64)); q = q[so]
    d, _ = cKDTree(tgt).query(q); b2 = d * d * (1 + 1e-5)
    t0 = time.time(); orders = {"hilbert": curve_order(tgt), "kd": kd_order(tgt)}; print("orders", time.time() - t0)
    qorders = {"hilbert": np.arange(len(q)), "kd": kd_order(q, 64)}
    rng = np.random.default_rng(0)
    for qn, qperm in qorders.items():
        qq = q[qperm]; bb = b2[qperm]
        G = len(qq) // 64
        gs = rng.choice(G, min(ngroups, G), replace=False)
        for tn, perm in orders.items():
            lo, hi = leaf_boxes(tgt, perm)
            tested = scanned = 0; vol = 0
            for g in gs:
                Q = qq[g * 64:(g + 1) * 64]; B = bb[g * 64:(g + 1) * 64]
                wl, wh = Q.min(0), Q.max(0); r2 = B.max()
                dd = np.maximum(np.maximum(lo - wh, wl - hi), 0); cand = (dd * dd).sum(-1) <= r2
                tested += cand.sum()
                D = box_d2(lo[cand], hi[cand], Q)
                scanned += (D <= B[:, None]).any(0).sum()
            print(f"{name}: queries {qn:8s} target {tn:8s}: per 64-query group leaves tested {tested/len(gs):6.1f} scanned {scanned/len(gs):6.1f}; mean leaf box diag {np.linalg.norm(hi-lo,axis=1).mean():.5f}")

if __name__ == "__main__":
    for name in sys.argv[1:]:
        simulate(name)
