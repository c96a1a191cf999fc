"""Independent numpy restatement of the per-point arithmetic of the hot path, used ONLY to
cross-check the C++ oracle (tests/test_oracle_kat.py).  TEST INFRASTRUCTURE.  PARITY UNPINNED.

Written from the reference source and SURVEY.md Appendix A, not from oracle/goicp_oracle.cpp:
fgoicp/registration.cu:27-60 (bounds), :258-278 (LUT nodes), :320-328 + CUDA linear filtering (lookup),
fgoicp/common.hpp:37-57 (rotation).  fp32 throughout; fma(a,b,c) is emulated as
float32(float64(a)*float64(b) + float64(c)) (exact product, one extra rounding in rare cases)."""
import numpy as np

f32 = np.float32
SQRT3 = f32(1.732050807568877)
PI = f32(3.141592653589793)


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def rot_apply(R, p):
    """device R*p for math-convention R (3,3) and points (n,3): fma(R[r,2], z, fma(R[r,1], y, R[r,0]*x))"""
    R = np.asarray(R, f32); p = np.asarray(p, f32)
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = fma(R[r, 2], p[:, 2], fma(R[r, 1], p[:, 1], (R[r, 0] * p[:, 0]).astype(f32)))
    return out


def dist_sq(a, b):
    d = (np.asarray(a, f32) - np.asarray(b, f32)).astype(f32)
    return fma(d[..., 2], d[..., 2], fma(d[..., 1], d[..., 1], (d[..., 0] * d[..., 0]).astype(f32)))


def lut_dims(bounds, res):
    b = np.asarray(bounds, f32).reshape(3, 2)
    return tuple(int(np.ceil(f32(f32(b[a, 1] - b[a, 0]) / f32(res)))) for a in range(3))


def lut_build(tgt, bounds, res):
    b = np.asarray(bounds, f32).reshape(3, 2)
    dx, dy, dz = lut_dims(bounds, res)
    pts = (np.asarray(tgt, f32) + (-b[:, 0])[None, :]).astype(f32)
    zz, yy, xx = np.meshgrid(np.arange(dz), np.arange(dy), np.arange(dx), indexing="ij")
    nodes = np.stack([xx, yy, zz], -1).astype(f32) * f32(res)
    out = np.full((dz, dy, dx), np.finfo(f32).max, f32)
    for p in pts:  # O(nodes * nt): small inputs only
        out = np.minimum(out, dist_sq(nodes, p[None, None, None, :]))
    return out


def _axis(u, dim, quant):
    ub = (u - f32(0.5)).astype(f32)
    fl = np.floor(ub)
    w = (ub - fl).astype(f32)
    if quant:
        w = (np.floor(w * f32(256) + f32(0.5)) / f32(256)).astype(f32)
    i = np.clip(fl, -1, dim).astype(np.int64)
    return np.clip(i, 0, dim - 1), np.clip(i + 1, 0, dim - 1), w


def lut_search(lut, bounds, res, q, quant=True):
    b = np.asarray(bounds, f32).reshape(3, 2)
    dz, dy, dx = lut.shape
    q = np.asarray(q, f32)
    scale = f32(1.0) / f32(res)
    u = [((q[:, a] + (-b[a, 0])).astype(f32) * scale).astype(f32) for a in range(3)]
    x0, x1, a = _axis(u[0], dx, quant)
    y0, y1, bb = _axis(u[1], dy, quant)
    z0, z1, c = _axis(u[2], dz, quant)
    lerp = lambda p, q_, w: fma(w, (q_ - p).astype(f32), p)
    c00 = lerp(lut[z0, y0, x0], lut[z0, y0, x1], a)
    c10 = lerp(lut[z0, y1, x0], lut[z0, y1, x1], a)
    c01 = lerp(lut[z1, y0, x0], lut[z1, y0, x1], a)
    c11 = lerp(lut[z1, y1, x0], lut[z1, y1, x1], a)
    return lerp(lerp(c00, c10, bb), lerp(c01, c11, bb), c)


def bounds(lut, lut_bounds, res, src, R, rot_span, tnodes4, fix_rot, quant=True, inliers=0):
    src = np.asarray(src, f32)
    rp = rot_apply(R, src)
    half_angle = f32(f32(f32(f32(rot_span) * SQRT3) * PI) / f32(2.0))
    sin_half = f32(np.sin(half_angle, dtype=np.float32))
    radius = fma(src[:, 2], src[:, 2], fma(src[:, 1], src[:, 1], (src[:, 0] * src[:, 0]).astype(f32)))
    lbs, ubs = [], []
    for tx, ty, tz, span in np.asarray(tnodes4, f32):
        q = (rp + np.array([tx, ty, tz], f32)[None, :]).astype(f32)
        d = np.sqrt(lut_search(lut, lut_bounds, res, q, quant)).astype(f32)
        if not fix_rot:
            d = (d - ((f32(2.0) * radius).astype(f32) * sin_half).astype(f32)).astype(f32)
        ub = np.where(d > 0, (d * d).astype(f32), f32(0))
        l = (d - f32(SQRT3 * f32(span))).astype(f32)
        lb = np.where(l > 0, (l * l).astype(f32), f32(0))
        if inliers and inliers < len(src):  # trimmed Go-ICP extension: the k smallest terms of each bound
            ub = np.sort(ub)[:inliers]
            lb = np.sort(lb)[:inliers]
        ubs.append(f32(ub.astype(np.float64).sum()))
        lbs.append(f32(lb.astype(np.float64).sum()))
    return np.array(lbs, f32), np.array(ubs, f32)


def point_distances(lut, lut_bounds, res, src, R, rot_span, tnode4, fix_rot, quant=True):
    """e_i = max(distance_i, 0) with `distance` of registration.cu:48-52 (the per-point quantity both bounds are functions of:
    ub_i = e_i^2, lb_i = max(e_i - sqrt3 * span, 0)^2) for ONE translation node."""
    src = np.asarray(src, f32)
    rp = rot_apply(R, src)
    tx, ty, tz, _ = np.asarray(tnode4, f32)
    q = (rp + np.array([tx, ty, tz], f32)[None, :]).astype(f32)
    d = np.sqrt(lut_search(lut, lut_bounds, res, q, quant)).astype(f32)
    if not fix_rot:
        half_angle = f32(f32(f32(f32(rot_span) * SQRT3) * PI) / f32(2.0))
        sin_half = f32(np.sin(half_angle, dtype=np.float32))
        radius = fma(src[:, 2], src[:, 2], fma(src[:, 1], src[:, 1], (src[:, 0] * src[:, 0]).astype(f32)))
        d = (d - ((f32(2.0) * radius).astype(f32) * sin_half).astype(f32)).astype(f32)
    return np.where(d > 0, d, f32(0)).astype(f32)


def rotation(x, y, z):
    """common.hpp:37-57 → (math-convention R, r, in_SO3)"""
    x, y, z = f32(x), f32(y), f32(z)
    r = f32(f32(f32(x * x) + f32(y * y)) + f32(z * z))
    if r > 1:
        return np.eye(3, dtype=f32), r, False
    ww = f32(1) - r
    w = np.sqrt(ww, dtype=f32)
    # standard unit-quaternion (w, x, y, z) rotation matrix, TRANSPOSED (glm fills columns)
    Rq = np.array([[ww + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), ww - x * x + y * y - z * z, 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), ww - x * x - y * y + z * z]], np.float64)
    return Rq.T.astype(f32), np.sqrt(r, dtype=f32), True


# ---------------------------------------------------------------------------------------------------------------------
# Eigen::JacobiSVD<Matrix3d> (fgoicp/icp3d.cu:118-121) — third statement of the published algorithm (Eigen 3.3 / 3.4:
# JacobiSVD::compute, real_2x2_jacobi_svd, JacobiRotation::makeJacobi), in matrix form: every plane rotation is applied as
# a 3x3 Givens matrix product, so rounding differs from the scalar forms in oracle/goicp_oracle.cpp and csrc/host/math3.hpp
# in the last bits only.  Used to cross-check what those two return on rank-deficient input, where the null-space columns
# are the algorithm's choice (pair order (1,0), (2,0), (2,1); the 2x2 block is taken with the larger index first).
# ---------------------------------------------------------------------------------------------------------------------
def _givens(p, q, c, s):
    """3x3 matrix that acts as [[c, s], [-s, c]] on coordinates (p, q)"""
    G = np.eye(3)
    G[p, p] = c; G[p, q] = s; G[q, p] = -s; G[q, q] = c
    return G


def jacobi_svd3(A):
    A = np.asarray(A, np.float64)
    tiny = np.finfo(np.float64).tiny
    scale = np.abs(A).max()
    if scale == 0.0:
        scale = 1.0
    W = A / scale
    U = np.eye(3); V = np.eye(3)
    maxdiag = np.abs(np.diag(W)).max()
    for _ in range(1000):
        finished = True
        for p in (1, 2):
            for q in range(p):
                thr = max(tiny, 2.0 * np.finfo(np.float64).eps * maxdiag)
                if not (abs(W[p, q]) > thr or abs(W[q, p]) > thr):
                    continue
                finished = False
                m = np.array([[W[p, p], W[p, q]], [W[q, p], W[q, q]]])
                t = m[0, 0] + m[1, 1]; d = m[1, 0] - m[0, 1]
                if abs(d) < tiny:
                    c1, s1 = 1.0, 0.0
                else:
                    u = t / d; h = np.sqrt(1.0 + u * u)
                    c1, s1 = u / h, 1.0 / h
                m = np.array([[c1, s1], [-s1, c1]]) @ m
                x, y, z = m[0, 0], m[0, 1], m[1, 1]
                if 2.0 * abs(y) < tiny:
                    cr, sr = 1.0, 0.0
                else:
                    tau = (x - z) / (2.0 * abs(y))
                    w = np.sqrt(tau * tau + 1.0)
                    tt = 1.0 / (tau + w) if tau > 0 else 1.0 / (tau - w)
                    n = 1.0 / np.sqrt(tt * tt + 1.0)
                    cr, sr = n, -np.sign(tt) * np.sign(y) * abs(tt) * n
                cl = c1 * cr + s1 * sr          # rot1 * right^T
                sl = -c1 * sr + s1 * cr
                L = _givens(p, q, cl, sl); Rt = _givens(p, q, cr, sr)
                W = L @ W @ Rt
                U = U @ L.T
                V = V @ Rt
                maxdiag = max(maxdiag, abs(W[p, p]), abs(W[q, q]))
        if finished:
            break
    S = np.abs(np.diag(W)).copy()
    for i in range(3):
        if W[i, i] < 0:
            U[:, i] = -U[:, i]
    S *= scale
    for i in range(3):
        pos = i + int(np.argmax(S[i:]))
        if S[pos] == 0.0:
            break
        if pos != i:
            S[[i, pos]] = S[[pos, i]]; U[:, [i, pos]] = U[:, [pos, i]]; V[:, [i, pos]] = V[:, [pos, i]]
    return U, S, V


def closest_orthogonal(H):
    """icp3d.cu:110-138 for the math-convention H (row = source axis, column = correspondence axis), in fp64"""
    U, S, V = jacobi_svd3(H)
    d = np.linalg.det(V @ U.T)
    return V @ np.diag([1.0, 1.0, d]) @ U.T
