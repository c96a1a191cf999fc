"""GPU: degenerate and tiny inputs through the C ABI — every case must give the oracle's answer or a
loud error, never a fault (the reference tests none of these; they are the domain's edge cases:
single points, duplicates, coplanar / collinear clouds, queries far outside the LUT, odd sizes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
f32 = np.float32
B3 = np.array([[-1, 1]] * 3, f32)


def _cmp_ops(fg, oracle, tgt, src, bounds, res, seed=0):
    hip = fg.Registration(tgt, src, bounds, res)
    orc = oracle.Registration(tgt, src, bounds, res)
    assert np.array_equal(hip.lut_read().view(np.uint32), orc.lut_get().view(np.uint32))
    rng = np.random.default_rng(seed)
    rn = fg.RotNode(0.2, -0.1, 0.3, 0.25)
    tn = np.concatenate([rng.uniform(-0.5, 0.5, (9, 3)), rng.choice([1.0, 0.25, 0.0625], (9, 1))], 1).astype(f32)
    for fix in (True, False):
        lb, ub = hip.compute_sse_error(rn, tn, fix)
        lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
        assert np.allclose(ub, ubo, rtol=1e-6, atol=1e-12) and np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * max(float(ubo.max()), 1e-12))
    R = fg.synth.random_rotation(rng, 40.0).astype(f32)
    t = rng.uniform(-0.2, 0.2, 3).astype(f32)
    a, b = float(hip.compute_sse_error(R, t)), float(orc.compute_sse_error(R, t))
    assert a == pytest.approx(b, rel=1e-6, abs=1e-12)
    w = (src @ R.T + t).astype(f32)
    Rh, th, cen, ABt, idx = hip.procrustes(w)
    Ro, to, ceno, ABto, idxo = orc.procrustes(w)
    assert np.array_equal(idx, idxo)
    assert np.allclose(cen, ceno, rtol=1e-6, atol=1e-7)
    sse, Ri, ti = fg.IterativeClosestPoint3D(hip, None, None, 20, 0.005, R, t).run()
    sse_o, Ri_o, ti_o, it_o = orc.icp(R, t, 20, 0.005)
    assert float(sse) == pytest.approx(float(sse_o), rel=1e-5, abs=1e-10)
    hip.close()
    return Rh, Ro


@pytest.mark.parametrize("nt,ns", [(1, 1), (2, 3), (3, 1), (33, 65), (257, 511), (1000, 1)])
def test_tiny_and_odd_sizes(fg, oracle, gpu_required, nt, ns):
    rng = np.random.default_rng(nt * 1000 + ns)
    tgt = rng.uniform(-0.8, 0.8, (nt, 3)).astype(f32)
    src = rng.uniform(-0.8, 0.8, (ns, 3)).astype(f32)
    _cmp_ops(fg, oracle, tgt, src, B3, 0.25)


def test_duplicate_points_and_exact_ties(fg, oracle, gpu_required):
    rng = np.random.default_rng(3)
    base = rng.uniform(-0.7, 0.7, (40, 3)).astype(f32)
    tgt = np.concatenate([base, base, base[::-1]])          # every target three times
    src = np.concatenate([base[:20], base[:20]])            # sources sit exactly ON targets, twice each
    _cmp_ops(fg, oracle, tgt, src, B3, 0.2)


def test_correspondence_tie_rule_compares_square_roots(fg, gpu_required):
    """The hand-derived case of tests/test_oracle_kat.py on the device, scan and brute-force kernels: squared distances one ulp
    apart with equal fp32 square roots tie, the first index wins (icp3d.cu:20-25); the SSE takes the smaller square."""
    from oracle import np_restatement as npr
    from tests.test_oracle_kat import sqrt_tie_pair
    near, far = sqrt_tie_pair()
    q = np.zeros((1, 3), f32)
    for flags in (0, fg.FLAG_BRUTE_FORCE_NN):
        for tgt, want in ((np.stack([far, near]), 0), (np.stack([near, far]), 0), (np.stack([far * 2, far, near]), 1)):
            reg = fg.Registration(tgt.astype(f32), q, B3, 0.5, flags=flags)
            assert reg.procrustes(q)[-1][0] == want
            assert reg.compute_sse_error(np.eye(3, dtype=f32), np.zeros(3, f32)) == npr.dist_sq(near[None, :], q)[0]
            reg.close()


def test_coplanar_and_collinear_clouds(fg, oracle, gpu_required):
    """Rank-deficient cross-covariance: a proper rotation, and the SAME one as the oracle's — both sides follow Eigen's JacobiSVD
    algorithm (icp3d.cu:118-121), whose null-space vectors decide R here (tests/test_gpu_rank_deficient.py has the ICP runs)."""
    rng = np.random.default_rng(4)
    plane = np.concatenate([rng.uniform(-0.8, 0.8, (200, 2)), np.full((200, 1), 0.1)], 1).astype(f32)
    bounds = np.array([[-1, 1], [-1, 1], [-0.2, 0.4]], f32)
    Rh, Ro = _cmp_ops(fg, oracle, plane, plane[:150].copy(), bounds, 0.1)
    assert np.allclose(Rh, Ro, atol=1e-5)
    for R in (Rh, Ro):
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-4) and np.linalg.det(R.astype(np.float64)) == pytest.approx(1.0, abs=1e-4)
    line = np.stack([np.linspace(-0.8, 0.8, 120), np.full(120, 0.05), np.full(120, -0.05)], 1).astype(f32)
    bounds = np.array([[-1, 1], [-0.2, 0.2], [-0.2, 0.2]], f32)
    Rh, Ro = _cmp_ops(fg, oracle, line, line[10:90].copy(), bounds, 0.05)
    assert np.allclose(Rh, Ro, atol=1e-5)
    assert np.allclose(Rh @ Rh.T, np.eye(3), atol=1e-4) and np.linalg.det(Rh.astype(np.float64)) == pytest.approx(1.0, abs=1e-4)


def test_queries_far_outside_the_lut(fg, oracle, gpu_required):
    rng = np.random.default_rng(5)
    tgt = rng.uniform(-0.3, 0.3, (500, 3)).astype(f32)
    src = rng.uniform(-0.3, 0.3, (300, 3)).astype(f32)
    bounds = np.array([[-0.3, 0.3]] * 3, f32)
    hip = fg.Registration(tgt, src, bounds, 0.05)
    orc = oracle.Registration(tgt, src, bounds, 0.05)
    for shift in (5.0, 1e3, 1e6):
        t = np.array([shift, -shift, 0.5 * shift], f32)
        a, b = float(hip.compute_sse_error(np.eye(3, dtype=f32), t)), float(orc.compute_sse_error(np.eye(3, dtype=f32), t))
        assert a == pytest.approx(b, rel=1e-6)
        tn = np.array([[shift, 0, 0, 0.5], [0, -shift, 0, 0.0625]], f32)
        lb, ub = hip.compute_sse_error(fg.RotNode(0, 0, 0, 0.5), tn, True)
        lbo, ubo = orc.compute_bounds(np.eye(3, dtype=f32), 0.5, tn, True)
        assert np.allclose(ub, ubo, rtol=1e-6) and np.allclose(lb, lbo, rtol=1e-6)
    hip.close()


def test_degenerate_inputs_fail_loudly(fg, gpu_required):
    pts = np.random.default_rng(6).uniform(-1, 1, (8, 3)).astype(f32)
    with pytest.raises(fg.FgoicpError):   # zero-extent axis: LUT dimension 0 (the reference would allocate an empty texture)
        fg.Registration(pts, pts, np.array([[-1, 1], [0.5, 0.5], [-1, 1]], f32), 0.1)
    with pytest.raises(fg.FgoicpError):
        fg.Registration(pts[:0], pts, B3, 0.1)
    with pytest.raises(fg.FgoicpError):
        fg.Registration(pts, pts, B3, 0.0)
    with pytest.raises(fg.FgoicpError):   # absurd resolution: dims beyond the supported range
        fg.Registration(pts, pts, B3, 1e-5)


def test_full_run_on_tiny_and_identical_clouds(fg, oracle, gpu_required):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-0.5, 0.5, (64, 3)).astype(f32)
    for sched in (fg.SCHEDULE_SERIAL, fg.SCHEDULE_ROUND):
        s = fg.FastGoICP(pts, pts.copy(), 0.05, 1e-3, schedule=sched, round_width=2)
        R, t = s.run()
        assert np.allclose(R, np.eye(3), atol=1e-5) and np.allclose(t, 0, atol=1e-5) and float(s.get_best_error()) < 1e-8
        st = s.stats()
        assert st["rot_cubes"] == 0 and st["icp_runs"] == 2   # initial ICP solves it; the BnB stops at the root (fgoicp.cpp:44)
        s.close()
    o = oracle.FastGoICP(pts, pts.copy(), 0.05, 1e-3).run()
    assert o["stats"]["rot_cubes"] == 0


def test_getters_can_be_polled_while_the_search_runs(fg, gpu_required):
    """The reference's external viewer polls get_best_error / get_best_transform / get_last_transform from another thread
    while run() is busy (README.md:19, fgoicp.hpp:31-43; upstream reads them racily).  Here they take a mutex inside the
    library: every polled transform is a proper rotation, the best error never increases, and the run is not disturbed."""
    import threading
    tgt, src, R_gt, t_gt = fg.synth.workload("small", angle_deg=150.0, min_angle_deg=110.0)
    ref = fg.FastGoICP(tgt, src, 0.02, 2e-5, schedule=fg.SCHEDULE_ROUND, round_width=0)
    R0, t0 = ref.run()
    e0, st0 = float(ref.get_best_error()), ref.stats()
    ref.close()
    s = fg.FastGoICP(tgt, src, 0.02, 2e-5, schedule=fg.SCHEDULE_ROUND, round_width=0)
    seen, stop, bad = [], threading.Event(), []

    def poll():
        while not stop.is_set():
            e = float(s.get_best_error())
            for R, t in (s.get_best_transform(), s.get_last_transform()):
                if not (np.allclose(R @ R.T, np.eye(3), atol=1e-4) and np.isfinite(t).all()):
                    bad.append(R)
            seen.append(e)

    th = threading.Thread(target=poll)
    th.start()
    R1, t1 = s.run()
    stop.set()
    th.join()
    assert not bad and len(seen) > 10
    during = [e for e in seen if e < 1e9]
    assert all(b <= a * (1 + 1e-6) for a, b in zip(during, during[1:]))  # the incumbent only improves
    assert np.array_equal(R0, R1) and np.array_equal(t0, t1) and float(s.get_best_error()) == e0
    assert s.stats()["trans_cubes"] == st0["trans_cubes"]
    s.close()


@pytest.mark.gpu
def test_non_finite_coordinates_are_refused(fg, gpu_required):
    """A NaN or an infinite coordinate in either cloud (the reference would carry it into every sum: NaN bounds, a search that never
    prunes) is an FGOICP_ERR_INVALID_ARG at construction, for the operator context and for the solver."""
    tgt, src, _, _ = fg.synth.workload("tiny", angle_deg=20.0)
    for bad_val in (np.nan, np.inf, -np.inf):
        for which in ("tgt", "src"):
            t, s = tgt.copy(), src.copy()
            (t if which == "tgt" else s)[7, 1] = bad_val
            with pytest.raises(fg.FgoicpError, match="non-finite"):
                fg.FastGoICP(t, s, 0.02, 1e-3)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    pcs = pcs.copy(); pcs[3, 0] = np.nan
    with pytest.raises(fg.FgoicpError, match="non-finite"):
        fg.Registration(pct, pcs, bounds, 0.02)


def test_struct_size_guards_the_extensible_structs(fg, gpu_required):
    """ABI 2: fgoicp_ctx_get_info writes no byte beyond the caller's struct_size, fgoicp_solver_set_exchange reads none — a
    caller compiled against a shorter struct neither gets overrun nor hands the library a garbage allgather_device."""
    import ctypes as C
    lib = fg._lib.load()
    tgt, src, _, _ = fg.synth.workload("tiny", angle_deg=20.0)
    s = fg.FastGoICP(tgt, src, 0.05, 1e-3)
    ctx = C.c_void_p(lib.fgoicp_solver_ctx(s._h))
    buf = (C.c_ubyte * 256)(*([0xAB] * 256))
    info = C.cast(buf, C.POINTER(fg._lib.CtxInfo))
    info.contents.struct_size = 0
    assert lib.fgoicp_ctx_get_info(ctx, info) == fg._lib.load().fgoicp_ctx_get_info(ctx, info) != 0  # refused: struct_size not set
    short = fg._lib.CtxInfo.lut_nodes.offset  # a caller that knows the struct up to lut_layout only
    info.contents.struct_size = short
    assert lib.fgoicp_ctx_get_info(ctx, info) == 0 and info.contents.lut_dims[0] > 0 and info.contents.lut_layout in (1, 2, 4)
    assert all(b == 0xAB for b in bytes(buf)[short:])  # nothing beyond the caller's struct was touched
    info.contents.struct_size = C.sizeof(fg._lib.CtxInfo)
    assert lib.fgoicp_ctx_get_info(ctx, info) == 0 and info.contents.points_per_item in (256, 512, 1024, 2048)
    calls = []
    ar = fg._lib.Exchange.ALLREDUCE_MIN(lambda b, n, u: calls.append("ar") or 0)
    ag = fg._lib.Exchange.ALLGATHER(lambda a, b, n, u: calls.append("ag") or 0)
    ex = fg._lib.Exchange(0, 1, ar, ag, None)
    ex.allgather_device = C.cast(0xDEAD0000, fg._lib.Exchange.ALLGATHER_DEVICE)  # garbage where an older caller's struct has already ended
    ex.struct_size = fg._lib.Exchange.allgather_device.offset
    assert lib.fgoicp_solver_set_exchange(s._h, C.byref(ex)) == 0
    R, t = s.run()  # world 1: no collective; above all nothing calls the garbage pointer
    ex.struct_size = 0
    assert lib.fgoicp_solver_set_exchange(s._h, C.byref(ex)) != 0 and b"struct_size" in lib.fgoicp_last_error()
    s.close()
