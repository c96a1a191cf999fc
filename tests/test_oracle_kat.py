"""Pins the CPU oracle: analytic known answers, reference conventions
and an independent numpy restatement.  The reference ships no golden vectors (parity unpinned);
these tests are what stands between the oracle and a silent misreading of the reference."""
import numpy as np
import pytest

from oracle import np_restatement as npr

f32 = np.float32


def ulp_diff(a, b):
    a = np.asarray(a, f32).view(np.int32).astype(np.int64)
    b = np.asarray(b, f32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


# ---- types / conventions ---------------------------------------------------------------
def test_rotation_is_transposed_quaternion_matrix(oracle):
    R, r, ok = oracle.rotation(0.5, 0.5, 0.5)  # w = 0.5: 120 deg about (1,1,1)
    assert ok and r == pytest.approx(np.sqrt(0.75), rel=1e-6)
    # glm::mat3(9 scalars) fills columns → the mathematical matrix is the transpose of the textbook one
    assert np.allclose(R, [[0, 1, 0], [0, 0, 1], [1, 0, 0]], atol=1e-6)
    for xyz in [(0.1, -0.2, 0.3), (0.7, 0.0, -0.7), (0, 0, 0)]:
        R, r, ok = oracle.rotation(*xyz)
        Rn, rn, okn = npr.rotation(*xyz)
        assert ok == okn and np.allclose(R, Rn, atol=1e-6) and r == pytest.approx(rn, rel=1e-6)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-5) and np.linalg.det(R.astype(np.float64)) == pytest.approx(1.0, abs=1e-5)


def test_rotation_outside_ball_keeps_squared_norm_and_identity(oracle):
    R, r, ok = oracle.rotation(0.75, 0.75, 0.75)
    assert not ok and np.array_equal(R, np.eye(3, dtype=f32))
    assert r == pytest.approx(3 * 0.75 ** 2)  # squared norm, common.hpp:42 early return


def test_overlaps_so3_formula(oracle):
    for x, y, z, s in [(0.5, 0.5, 0.5, 0.5), (0.75, 0.75, 0.75, 0.25), (0.875, 0.875, 0.875, 0.125), (0.9375, 0.9375, 0.0625, 0.0625),
                       (0.25, 0.25, 0.25, 0.25)]:
        _, r, _ = oracle.rotation(x, y, z)
        expect = f32(r) - f32(2 * s * (abs(x) + abs(y) + abs(z))) + f32(3 * s * s) <= 1
        assert oracle.rotnode_overlaps(x, y, z, s) == bool(expect)
    assert oracle.rotnode_overlaps(0.75, 0.75, 0.75, 0.25)        # corner cube that still touches the ball
    assert not oracle.rotnode_overlaps(0.9375, 0.9375, 0.9375, 0.0625)


def test_priority_order_smallest_lb_then_largest_span(oracle):
    lb = np.array([5, 1, 1, 3, 0.5], f32)
    span = np.array([1, 0.25, 0.5, 0.125, 0.0625], f32)
    order = oracle.transnode_pop_order(lb, span)
    assert list(order) == [4, 2, 1, 3, 0]


# ---- LUT ------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_lut_case(oracle):
    rng = np.random.default_rng(3)
    tgt = rng.uniform(-0.5, 0.45, size=(40, 3)).astype(f32)
    src = rng.uniform(-0.4, 0.4, size=(64, 3)).astype(f32)
    bounds = np.array([[tgt[:, k].min(), tgt[:, k].max()] for k in range(3)], f32)
    res = 0.07
    reg = oracle.Registration(tgt, src, bounds, res)
    return tgt, src, bounds, res, reg


def test_lut_dims_and_nodes_match_numpy(small_lut_case):
    tgt, src, bounds, res, reg = small_lut_case
    assert reg.lut_dims() == npr.lut_dims(bounds, res)
    a, b = reg.lut_get(), npr.lut_build(tgt, bounds, res)
    d = ulp_diff(a, b)
    assert d.max() <= 1 and (d == 0).mean() > 0.999


def test_lut_single_point_analytic(oracle):
    tgt = np.array([[0.2, -0.1, 0.3], [0.2, -0.1, 0.3]], f32)  # degenerate AABB would give dims 0: widen it
    bounds = np.array([[0.0, 0.5], [-0.4, 0.1], [0.0, 0.6]], f32)
    reg = oracle.Registration(tgt, tgt, bounds, 0.1)
    dx, dy, dz = reg.lut_dims()
    assert (dx, dy, dz) == (5, 5, 6)
    lut = reg.lut_get()
    for (i, j, k) in [(0, 0, 0), (2, 3, 3), (4, 4, 5)]:
        node = np.array([i, j, k], np.float64) * 0.1
        want = np.sum((node - (tgt[0].astype(np.float64) - bounds[:, 0])) ** 2)
        assert lut[k, j, i] == pytest.approx(want, rel=1e-5)  # index (z*dy + y)*dx + x


def test_lut_search_cuda_filtering_semantics(oracle):
    """Linear field T[i,j,k] = 2i + 3j - k + 7 is reproduced by trilinear filtering; the sample
    position is u - 0.5 (texel centres), indices clamp."""
    bounds = np.array([[0, 1], [0, 1], [0, 1]], f32)
    res = 0.125  # dims 8^3, scale 8
    reg = oracle.Registration(np.zeros((1, 3), f32), np.zeros((1, 3), f32), bounds, res, build_lut=False, quantize=False)
    assert reg.lut_dims() == (8, 8, 8)
    k, j, i = np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij")
    T = (2 * i + 3 * j - k + 7).astype(f32)
    reg.lut_set(T)
    rng = np.random.default_rng(0)
    q = rng.uniform(0.07, 0.93, size=(200, 3)).astype(f32)  # interior: u - 0.5 in [0.06, 6.94]
    u = q.astype(np.float64) * 8 - 0.5
    want = 2 * u[:, 0] + 3 * u[:, 1] - u[:, 2] + 7
    assert np.allclose(reg.lut_search(q), want, rtol=1e-5, atol=1e-5)
    # exactly at a texel centre ((i + 0.5) * res) the value is the texel itself
    assert reg.lut_search(np.array([[(3 + 0.5) * res, (1 + 0.5) * res, (6 + 0.5) * res]], f32))[0] == T[6, 1, 3]
    # clamp addressing
    assert reg.lut_search(np.array([[-5, -5, -5]], f32))[0] == T[0, 0, 0]
    assert reg.lut_search(np.array([[9, 9, 9]], f32))[0] == T[7, 7, 7]
    assert reg.lut_search(np.array([[9, -9, 0.0625 * 5]], f32))[0] == pytest.approx(T[2, 0, 7], abs=1e-5)


def test_lut_search_weight_quantisation(oracle):
    bounds = np.array([[0, 1], [0, 1], [0, 1]], f32)
    reg = oracle.Registration(np.zeros((1, 3), f32), np.zeros((1, 3), f32), bounds, 0.25, build_lut=False, quantize=True)
    k, j, i = np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij")
    reg.lut_set(i.astype(f32))  # T = x index
    # u_x = 4*q_x; u - 0.5 = 1.3 → alpha = 0.3 → 1.8 fixed point: round(76.8)/256 = 77/256
    q = np.array([[1.8 / 4, 0.5, 0.5]], f32)
    assert reg.lut_search(q)[0] == pytest.approx(1 + 77 / 256, abs=1e-6)


def test_lut_search_matches_numpy(small_lut_case):
    tgt, src, bounds, res, reg = small_lut_case
    rng = np.random.default_rng(1)
    q = rng.uniform(-1.0, 1.0, size=(4000, 3)).astype(f32)
    for quant in (True, False):
        reg2 = reg if quant else None
        if not quant:
            from oracle import pyoracle
            reg2 = pyoracle.Registration(tgt, src, bounds, res, quantize=False)
        a = reg2.lut_search(q)
        b = npr.lut_search(reg2.lut_get(), bounds, res, q, quant)
        d = ulp_diff(a, b)
        assert d.max() <= 2 and (d == 0).mean() > 0.995


# ---- bounds -------------------------------------------------------------------------------------
@pytest.mark.parametrize("fix_rot", [True, False])
def test_bounds_match_numpy(small_lut_case, oracle, fix_rot):
    tgt, src, bounds, res, reg = small_lut_case
    R, _, _ = oracle.rotation(0.2, -0.3, 0.1)
    rng = np.random.default_rng(2)
    tn = np.concatenate([rng.uniform(-0.3, 0.3, (9, 3)), rng.choice([0.5, 0.25, 0.0625], (9, 1))], 1).astype(f32)
    lb, ub = reg.compute_bounds(R, 0.25, tn, fix_rot)
    lbn, ubn = npr.bounds(reg.lut_get(), bounds, res, src, R, 0.25, tn, fix_rot)
    assert np.allclose(ub, ubn, rtol=2e-6) and np.allclose(lb, lbn, rtol=2e-6, atol=1e-6)
    assert np.all(lb <= ub)


def test_bounds_single_point_analytic(oracle):
    """One source point, smooth LUT region: ub = max(d - gamma, 0)^2, lb = max(d - gamma - sqrt3*span, 0)^2."""
    tgt = np.array([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]], f32)
    bounds = np.array([[0, 1], [0, 1], [0, 1]], f32)
    src = np.array([[0.3, 0.2, 0.1]], f32)
    reg = oracle.Registration(tgt, src, bounds, 0.02)
    t = np.array([[0.1, 0.15, 0.2, 0.125]], f32)
    q = src[0] + t[0, :3]
    dsq = reg.lut_search(q[None, :])[0]
    d = np.sqrt(np.float64(dsq))
    lb, ub = reg.compute_bounds(np.eye(3), 0.25, t, True)
    assert ub[0] == pytest.approx(d * d, rel=1e-6)
    assert lb[0] == pytest.approx(max(d - np.sqrt(3) * 0.125, 0) ** 2, rel=1e-5)
    # the LUT value approximates the squared distance at q - res/2 (half-voxel shift)
    assert dsq == pytest.approx(np.sum((q.astype(np.float64) - 0.01) ** 2), rel=2e-2)
    lb2, ub2 = reg.compute_bounds(np.eye(3), 0.25, t, False)
    gamma = 2 * float(np.sum(src[0].astype(np.float64) ** 2)) * np.sin(0.25 * np.sqrt(3) * np.pi / 2)  # squared norm: reference quirk
    assert ub2[0] == pytest.approx(max(d - gamma, 0) ** 2, rel=1e-5)
    assert lb2[0] == pytest.approx(max(d - gamma - np.sqrt(3) * 0.125, 0) ** 2, rel=1e-4, abs=1e-9)


# ---- exact SSE / Procrustes / ICP -----------------------------------------------------------------
def test_exact_sse_known_motion(oracle, fg):
    rng = np.random.default_rng(4)
    tgt = rng.uniform(-1, 1, (300, 3)).astype(f32)
    R = fg.synth.random_rotation(rng, 40.0)
    t = np.array([0.1, -0.2, 0.05])
    src = ((tgt.astype(np.float64) - t) @ R).astype(f32)  # R @ src + t == tgt
    reg = oracle.Registration(tgt, src, np.array([[-1, 1]] * 3, f32), 0.5, build_lut=False)
    assert float(reg.compute_sse_error(R.astype(f32), t.astype(f32))) < 1e-9
    brute = np.min(((src[:, None, :].astype(np.float64) - tgt[None, :, :]) ** 2).sum(-1), axis=1).sum()
    assert float(reg.compute_sse_error(np.eye(3, dtype=f32), np.zeros(3, f32))) == pytest.approx(brute, rel=1e-5)


def test_closest_orthogonal_matches_numpy_svd(oracle):
    rng = np.random.default_rng(5)
    for trial in range(20):
        H = rng.normal(size=(3, 3))
        if trial % 5 == 0:
            H[:, 2] = H[:, 0] * 0.5 + H[:, 1] * 0.25  # rank 2
        U, S, Vt = np.linalg.svd(H)
        V = Vt.T
        D = np.diag([1, 1, np.linalg.det(V @ U.T)])
        want = V @ D @ U.T
        ABt = H.T.reshape(9).astype(f32)  # glm ABt[c][r] = H(r, c): flat[c*3 + r]
        got = oracle.closest_orthogonal(ABt).reshape(3, 3).T  # glm → math
        assert np.allclose(got, want, atol=2e-5), trial
        assert np.linalg.det(got.astype(np.float64)) == pytest.approx(1.0, abs=1e-4)
    U, S, V = oracle.svd3(H)
    assert np.allclose(U @ np.diag(S) @ V.T, H, atol=1e-12) and np.all(np.diff(S) <= 0)


def _rank_deficient_family(rng, trial):
    """3x3 cross-covariances of every rank, as fp32 values (what icp3d.cu:166 hands to the SVD)"""
    H = rng.normal(size=(3, 3)) * 10.0 ** rng.uniform(-9, 2)
    k = trial % 8
    if k == 1:
        H[:, 1] = 2 * H[:, 0]                                                   # rank 2, exactly
    if k == 2:
        H = np.outer(rng.normal(size=3), rng.normal(size=3))                    # rank 1 up to the fp32 rounding of its entries
    if k == 3:
        H = H @ np.diag([1, 1, -1])                                             # det < 0: needs diag(1, 1, det)
    if k == 4:
        H = np.zeros((3, 3))                                                    # every correspondence on one target point
    if k == 5:
        H = np.outer(rng.normal(size=3), rng.normal(size=3)) + 1e-9 * rng.normal(size=(3, 3))
    if k == 6:
        H = np.diag(rng.normal(size=3))
    if k == 7:
        H = np.outer(rng.normal(size=3), [1.0, 0.0, 0.0]) * 1e-12               # rank 1, tiny
    return H.astype(f32).astype(np.float64)


def test_svd_follows_eigens_jacobi_algorithm(oracle):
    """icp3d.cu:118-121 calls Eigen::JacobiSVD<Matrix3d>.  Known answers of THAT algorithm which a different SVD would not give
    (they decide R on rank-deficient input): no rotation is applied to a diagonal matrix, so U and V stay (signed, permuted) unit
    vectors; the sort swaps with the first maximum of the tail; a negative diagonal entry negates U's column, not V's; the zero
    matrix returns identities.  Then the scalar restatement against the numpy matrix-form one on every rank."""
    from oracle import np_restatement as npr
    I = np.eye(3)
    U, S, V = oracle.svd3(np.zeros((3, 3)))
    assert np.array_equal(U, I) and np.array_equal(V, I) and np.array_equal(S, [0, 0, 0])
    assert np.array_equal(oracle.closest_orthogonal(np.zeros(9, f32)).reshape(3, 3), I)      # R = I, not a noise rotation
    U, S, V = oracle.svd3(np.diag([3.0, 2.0, 1.0]))
    assert np.array_equal(U, I) and np.array_equal(V, I) and np.array_equal(S, [3, 2, 1])
    U, S, V = oracle.svd3(np.diag([1.0, 2.0, 3.0]))     # i=0 swaps with column 2 (the maximum); then (2, 1) is in order
    assert np.array_equal(S, [3, 2, 1]) and np.array_equal(U, I[:, [2, 1, 0]]) and np.array_equal(V, I[:, [2, 1, 0]])
    U, S, V = oracle.svd3(np.diag([2.0, 2.0, 5.0]))     # equal values: the FIRST maximum of the tail is taken -> no second swap
    assert np.array_equal(S, [5, 2, 2]) and np.array_equal(U, I[:, [2, 1, 0]])
    U, S, V = oracle.svd3(np.diag([2.0, -5.0, 1.0]))
    assert np.array_equal(S, [5, 2, 1]) and np.array_equal(V, I[:, [1, 0, 2]]) and np.array_equal(U, np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1.0]]))
    U, S, V = oracle.svd3(np.diag([4.0, 0.0, 0.0]))     # the sort stops at the first zero
    assert np.array_equal(U, I) and np.array_equal(V, I)
    # one 2x2 block, by hand: H = [[0, 1], [-1, 0]] (+) 1 is a rotation by -90 deg: d = m10 - m01 with the LARGER index first
    H = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    U, S, V = oracle.svd3(H)
    assert np.allclose(U @ np.diag(S) @ V.T, H, atol=1e-15) and np.allclose(S, 1)
    R = oracle.closest_orthogonal(H.T.reshape(9).astype(f32)).reshape(3, 3).T
    assert np.allclose(R, H.T, atol=1e-7)               # R = V U^T = H^T for an orthogonal H
    rng = np.random.default_rng(12)
    for trial in range(400):
        H = _rank_deficient_family(rng, trial)
        U, S, V = oracle.svd3(H)
        scale = max(np.abs(H).max(), 1e-300)
        assert np.allclose(U @ np.diag(S) @ V.T, H, atol=1e-14 * scale) and np.all(np.diff(S) <= 0) and np.all(S >= 0)
        assert np.allclose(U.T @ U, I, atol=1e-13) and np.allclose(V.T @ V, I, atol=1e-13)
        R = oracle.closest_orthogonal(H.T.reshape(9).astype(f32)).reshape(3, 3).T
        if trial % 8 != 7:  # exact zero columns: the scalar form keeps exact zeros where the matrix form leaves 1e-17 noise, and the
            #                   sign of that noise picks the null-space vectors — only a scalar statement can follow Eigen there
            Rn = npr.closest_orthogonal(H)
            assert np.allclose(R, Rn, atol=2e-7), (trial, R, Rn)  # the same member of the solution family, on every other rank
        assert np.linalg.det(R.astype(np.float64)) == pytest.approx(1.0, abs=1e-5)


def test_svd_choice_only_matters_on_rank_deficient_input(oracle):
    """The convention flip `svd_r2_two_sided` (round 2's own Jacobi): same R on full-rank H, a different member of the family on
    rank-deficient H — which is why both sides now restate Eigen's algorithm (fuzz seed 7, cases 103 and 505)."""
    rng = np.random.default_rng(13)
    differs = 0
    try:
        for trial in range(200):
            H = _rank_deficient_family(rng, trial)
            ABt = H.T.reshape(9).astype(f32)
            a = oracle.closest_orthogonal(ABt)
            oracle.set_conventions(svd_r2_two_sided=1)
            b = oracle.closest_orthogonal(ABt)
            oracle.reset_conventions()
            S = np.linalg.svd(H, compute_uv=False)
            if S[0] > 0 and S[1] > 1e-3 * S[0]:      # rank >= 2 and well conditioned: R is unique
                assert np.allclose(a, b, atol=1e-4 if S[2] < 1e-6 * S[0] else 2e-6), trial
            elif not np.allclose(a, b, atol=1e-3):
                differs += 1
    finally:
        oracle.reset_conventions()
    assert differs > 10


def test_procrustes_recovers_known_motion(oracle, fg):
    rng = np.random.default_rng(6)
    tgt = rng.uniform(-1, 1, (400, 3)).astype(f32)
    R0 = fg.synth.random_rotation(rng, 0.5)  # small: every nearest neighbour is the true correspondence
    t0 = np.array([0.002, -0.001, 0.0005])
    work = ((tgt[:250].astype(np.float64) - t0) @ R0).astype(f32)
    reg = oracle.Registration(tgt, work, np.array([[-1, 1]] * 3, f32), 0.5, build_lut=False)
    R, t, cen, ABt, idx = reg.procrustes(work)
    assert np.array_equal(idx, np.arange(250))
    assert np.allclose(R, R0, atol=1e-5) and np.allclose(t, t0, atol=1e-5)
    assert np.allclose(cen[:3], work.mean(0), atol=1e-6) and np.allclose(cen[3:], tgt[:250].mean(0), atol=1e-6)


def test_icp_converges_and_respects_loop_rules(oracle, fg):
    tgt, src, R_gt, t_gt = fg.synth.workload("tiny", angle_deg=8.0)
    pct, pcs, *_ , bounds = fg.synth.preprocess(tgt, src)
    reg = oracle.Registration(pct, pcs, bounds, 0.2, build_lut=False)
    sse0 = reg.compute_sse_error(np.eye(3, dtype=f32), np.zeros(3, f32))
    sse, R, t, iters = reg.icp(np.eye(3, dtype=f32), np.zeros(3, f32), 100, 0.005)
    assert 1 <= iters <= 100 and sse < sse0
    ang = np.degrees(np.arccos(np.clip((np.trace(R.astype(np.float64).T @ R_gt) - 1) / 2, -1, 1)))
    assert ang < 3.0  # partial overlap biases plain ICP a little; it must land in the right basin
    # the returned sse is the exact SSE of the returned motion
    assert float(reg.compute_sse_error(R, t)) == pytest.approx(float(sse), rel=1e-6)
    # max_iter = 1: one Procrustes step, result = better of (step, 1e10)
    sse1, *_ , it1 = reg.icp(np.eye(3, dtype=f32), np.zeros(3, f32), 1, 0.005)
    assert it1 == 1 and sse1 <= sse0 * 1.0001
    # max_iter = 0: loop never runs; (1e10 < 2e10) → returns (1e10, R0, t0)  (icp3d.cu:94, 106)
    sse_z, Rz, tz, itz = reg.icp(np.eye(3, dtype=f32), np.zeros(3, f32), 0, 0.005)
    assert itz == 0 and float(sse_z) == pytest.approx(1e10)


# ---- pre-processing and the full driver -------------------------------------------------------
def test_preprocessing(oracle, fg):
    tgt, src, *_ = fg.synth.workload("tiny", angle_deg=30.0)
    g = oracle.FastGoICP(tgt[:300], src[:200], 0.2, 1e-3)
    pp = g.preproc()
    assert np.allclose(pp["offset_pcs"], -src[:200].mean(0), atol=1e-6) and np.allclose(pp["offset_pct"], -tgt[:300].mean(0), atol=1e-6)
    s = 1.0 / np.abs(src[:200] - src[:200].mean(0)).max()
    assert float(pp["scale"]) == pytest.approx(s, rel=1e-5)
    assert np.abs(pp["pcs"]).max() == pytest.approx(1.0, rel=1e-6)  # source fits [-1, 1]^3 exactly
    assert np.allclose(pp["pct"], (tgt[:300] - tgt[:300].mean(0)) * s, atol=1e-5)
    assert np.allclose(pp["bounds"][:, 0], pp["pct"].min(0)) and np.allclose(pp["bounds"][:, 1], pp["pct"].max(0))
    # fg.synth.preprocess is the numpy twin used by the operator tests
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt[:300], src[:200])
    assert np.array_equal(pct, pp["pct"]) and np.array_equal(pcs, pp["pcs"]) and np.array_equal(bounds, pp["bounds"])


def test_full_run_recovers_known_se3(oracle, fg):
    """The same points under a known SE(3) (exact correspondences exist), rotation large enough
    that the initial ICP fails and the BnB has to find the basin: the ground truth comes back."""
    rng = np.random.default_rng(21)
    tgt, _, _, _ = fg.synth.make_pair(600, 10, (0.156, 0.152, 0.118), seed=21)
    R_gt = fg.synth.random_rotation(rng, 150.0, 140.0)
    t_gt = np.array([0.01, -0.02, 0.015])
    src = ((tgt[:400].astype(np.float64) - t_gt) @ R_gt).astype(f32)  # R_gt @ src + t_gt == tgt[:400]
    g = oracle.FastGoICP(tgt, src, 0.05, 1e-3)
    out = g.run()
    assert out["stats"]["rot_cubes"] > 0  # the initial ICP alone did not solve it
    ang = np.degrees(np.arccos(np.clip((np.trace(out["R"].astype(np.float64).T @ R_gt) - 1) / 2, -1, 1)))
    assert ang < 0.05 and np.linalg.norm(out["t"] - t_gt) < 1e-4
    assert float(out["best_sse"]) < 1e-6


@pytest.mark.parametrize("case", ["plain", "duplicates", "far_queries", "lattice_ties"])
def test_grid_nn_of_the_cpu_baseline_equals_the_brute_force_loops(oracle, fg, case):
    """bench.py's cpu_baseline leg searches nearest neighbours through a uniform grid (the stand-in for the nanoflann kd-tree the
    reference's README names): minimum squared distances and first-index correspondences are those of the restated O(n*m)
    loops (registration.cu:162-174, icp3d.cu:11-28), bit for bit — duplicates, queries far outside the target's box and exact
    ties on a lattice included."""
    tgt, src, *_ = fg.synth.make_pair(2500, 1500, (0.156, 0.152, 0.118), seed=3)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    R = fg.synth.random_rotation(np.random.default_rng(1), 40.0).astype(np.float32)
    t = np.array([0.05, -0.08, 0.02], np.float32)
    if case == "duplicates":
        pct = np.concatenate([pct, pct[:300]])
    elif case == "far_queries":
        pcs = (pcs * 3).astype(np.float32)
    elif case == "lattice_ties":
        pct = (np.round(pct * 8) / 8).astype(np.float32); pcs = (np.round(pcs * 8) / 8).astype(np.float32)
        R = np.eye(3, dtype=np.float32); t = np.zeros(3, np.float32)
    brute = oracle.Registration(pct, pcs, bounds, 0.1, build_lut=False)
    grid = oracle.Registration(pct, pcs, bounds, 0.1, build_lut=False)
    grid.use_grid(True)
    assert brute.compute_sse_error(R, t).view(np.uint32) == grid.compute_sse_error(R, t).view(np.uint32)
    w = (pcs @ R.T + t).astype(np.float32)
    assert np.array_equal(brute.procrustes(w)[-1], grid.procrustes(w)[-1])
    grid.set_inliers(len(pcs) // 2); brute.set_inliers(len(pcs) // 2)
    assert brute.compute_sse_error(R, t).view(np.uint32) == grid.compute_sse_error(R, t).view(np.uint32)


def sqrt_tie_pair():
    """Two points whose squared distances to the origin differ by one ulp while their fp32 square roots are equal (found by stepping
    one coordinate one ulp at a time; deterministic)."""
    rng = np.random.default_rng(0)
    d2 = lambda p: npr.dist_sq(p[None, :], np.zeros((1, 3), np.float32))[0]
    while True:
        a = rng.uniform(0.3, 0.6, 3).astype(np.float32)
        b = a.copy()
        b[0] = np.nextafter(b[0], np.float32(1))
        da, db = d2(a), d2(b)
        if db > da and np.sqrt(da, dtype=np.float32) == np.sqrt(db, dtype=np.float32):
            return a, b


def test_correspondence_tie_rule_compares_square_roots(oracle):
    """kernFindNearestNeighbor compares glm::distance (a square root) with a strict '>' (icp3d.cu:20-25): two targets whose
    SQUARED distances differ by one ulp but whose square roots round to the same float tie, and the first index wins — even when
    it is the (by one ulp) farther one.  kernComputeClosestError compares squared distances (registration.cu:162-174) and takes
    the smaller."""
    near, far = sqrt_tie_pair()
    q = np.zeros((1, 3), np.float32)
    bounds = np.array([[-1, 1]] * 3, np.float32)
    for tgt, want in ((np.stack([far, near]), 0), (np.stack([near, far]), 0), (np.stack([far * 2, far, near]), 1)):
        reg = oracle.Registration(tgt.astype(np.float32), q, bounds, 0.5, build_lut=False)
        assert reg.procrustes(q)[-1][0] == want
        sse = reg.compute_sse_error(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
        assert sse == npr.dist_sq(near[None, :], q)[0]
