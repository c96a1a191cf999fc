"""Full-size GPU tests of the BASELINE.json configs: every config that runs on one GPU runs here at its real
size, through the C ABI.

  (a) dragon shape, 437 645 x 437 645 points (configs[2]): full FastGoICP.run() under both schedules and both thresholds,
      bounds against the oracle and additive over a split of the source, exact-NN scan == brute force on the whole cloud;
  (b) 1 M points, 20 % uniform outliers, trimmed Go-ICP (configs[4]): trimmed bounds == numpy.partition of the per-point
      values read back from the device == the oracle, device-wide selection == one-block selection, pruned exact NN ==
      unpruned, full run recovers the ground truth;
  (c) the test/bunny.toml shape at lut_resolution 0.002 (configs[0]'s LUT: ~920 x 900 x 700 nodes, 9.5 GB packed): LUT nodes
      at sampled indices == single-node brute force, lookups bit-exact on those cells, default-threshold runs.

The oracle cannot build a LUT of these sizes (O(nodes * nt) on the CPU), so where the oracle's operators are used its LUT is
filled from the device LUT, and the device LUT itself is pinned by brute-force node values at sampled indices.
Size-independent properties carry the rest (additivity, schedule independence, ground truth, A/B of two device paths)."""
import numpy as np
import pytest

from oracle import np_restatement as npr

pytestmark = pytest.mark.gpu
f32 = np.float32
SQRT3 = f32(1.732050807568877)


def ang_deg(R, R_gt):
    return float(np.degrees(np.arccos(np.clip((np.trace(np.asarray(R, np.float64).T @ R_gt) - 1) / 2, -1, 1))))


def tnodes(rng, B, span, lim=0.5):
    return np.concatenate([rng.uniform(-lim, lim, (B, 3)), np.full((B, 1), span)], axis=1).astype(f32)


def brute_nodes(tgt, bounds, res, xyz):
    """buildLUTKernel for single nodes (registration.cu:258-278): min_j |node * res - (tgt_j - min_bound)|^2 in fp32."""
    pts = (tgt + (-np.asarray(bounds, f32)[:, 0])[None, :]).astype(f32)
    out = np.empty(len(xyz), f32)
    step = max(1, (1 << 24) // len(pts))
    for a in range(0, len(xyz), step):
        nodes = (xyz[a:a + step].astype(f32) * f32(res)).astype(f32)
        out[a:a + step] = npr.dist_sq(nodes[:, None, :], pts[None, :, :]).min(axis=1)
    return out


def same_result(a, b, rel=1e-5):
    (Ra, ta, ea), (Rb, tb, eb) = a, b
    return (abs(float(ea) - float(eb)) <= rel * max(abs(float(eb)), 1e-12) and np.allclose(Ra, Rb, atol=1e-5)
            and np.allclose(ta, tb, atol=1e-5 * max(1.0, float(np.abs(tb).max()))))


# ------------------------------------------------------------------------------------------------
# (a) dragon shape, 437 645 x 437 645
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dragon(fg, gpu_required):
    tgt, src, R_gt, t_gt = fg.synth.workload("dragon", angle_deg=150.0, min_angle_deg=110.0)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    return dict(tgt=tgt, src=src, R_gt=R_gt, t_gt=t_gt, pct=pct, pcs=pcs, bounds=bounds)


@pytest.mark.parametrize("mse", [1e-3, 5e-6])
def test_dragon_full_run_is_schedule_independent_and_finds_the_ground_truth(fg, dragon, mse):
    """The whole 437k pair through FastGoICP::run() (fgoicp.cpp:10-30): the reference's order (SERIAL) and expansion rounds
    end in the same optimum; 5e-6 is the certify regime (ns * mse below the residual), 1e-3 the reference's default."""
    res = {}
    for name, sched, K in (("serial", fg.SCHEDULE_SERIAL, 1), ("round", fg.SCHEDULE_ROUND, 0)):
        s = fg.FastGoICP(dragon["tgt"], dragon["src"], 0.005, mse, schedule=sched, round_width=K)
        R, t = s.run()
        res[name] = (R, t, float(s.get_best_error()))
        st = s.stats()
        assert st["trans_cubes"] > 0 and st["icp_runs"] >= 2
        assert s.registration.sort_fallbacks()[1] == 0  # every tick's locality sort was a permutation (checked on the device)
        s.close()
    assert same_result(res["serial"], res["round"])
    R, t, _ = res["serial"]
    assert ang_deg(R, dragon["R_gt"]) < 0.1 and np.linalg.norm(t - dragon["t_gt"]) < 1e-3 * 0.22


def test_dragon_sharded_over_two_ranks(fg, dragon):
    """BASELINE configs[3] (the dragon pair, the outer BnB sharded over ranks) on the one GPU of a test box: two in-process ranks on device 0
    through fgoicp_multi, certify regime (ns * mse below the residual).
    SERIAL — the reference's trajectory (fgoicp.cpp:32-100), the inner BnBs of every speculative evaluation dealt over the ranks: EVERY
    rank must end with the one-GPU SERIAL run's counters and its (R, t, sse) bit for bit; one rank replayed alone against the recording
    ends in the same state; the recorded object is destroyed afterwards (the sequence whose teardown once aborted).
    ROUND — north_star's partitioning (a round's children dealt over the ranks; at this size the cooperative flow: bounds exchanged
    first, triggers in the one-GPU child order, every ICP run by all ranks): the one-GPU optimum to 1e-5, identical incumbents on both ranks,
    the rotation cubes split between them."""
    mse = 5e-6
    keys = ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb", "rounds")
    one = fg.FastGoICP(dragon["tgt"], dragon["src"], 0.005, mse, schedule=fg.SCHEDULE_SERIAL)
    R1, t1 = one.run()
    e1, st1 = one.get_best_error(), one.stats()
    one.close()
    m = fg.MultiGoICP(dragon["tgt"], dragon["src"], 0.005, mse, devices=[0, 0], transport=fg.TRANSPORT_IN_PROCESS, schedule=fg.SCHEDULE_SERIAL)
    m.set_record(True)
    R, t = m.run()
    assert np.array_equal(R, R1) and np.array_equal(t, t1) and f32(m.get_best_error()).view(np.uint32) == f32(e1).view(np.uint32)
    for r in range(2):
        st = m.stats(r)
        assert [st[k] for k in keys] == [st1[k] for k in keys], (r, st, st1)
        assert m.registration(r).sort_fallbacks()[1] == 0
    assert m.recorded(0)[0] > 0
    assert m.replay_rank(1) > 0
    st = m.stats(1)
    assert [st[k] for k in keys] == [st1[k] for k in keys] and m.get_best_error(1) == m.get_best_error(0)
    m.close()
    one = fg.FastGoICP(dragon["tgt"], dragon["src"], 0.005, mse, schedule=fg.SCHEDULE_ROUND, round_width=0)
    Rr, tr = one.run()
    er, str1 = one.get_best_error(), one.stats()
    one.close()
    assert same_result((Rr, tr, er), (R1, t1, e1))
    m = fg.MultiGoICP(dragon["tgt"], dragon["src"], 0.005, mse, devices=[0, 0], transport=fg.TRANSPORT_IN_PROCESS, schedule=fg.SCHEDULE_ROUND, round_width=0)
    R, t = m.run()
    assert same_result((R, t, m.get_best_error()), (Rr, tr, er))
    sts = [m.stats(r) for r in range(2)]
    assert m.get_best_error(1) == m.get_best_error(0)
    assert min(s["rot_cubes"] for s in sts) > 0.3 * str1["rot_cubes"] and 0.7 * str1["trans_cubes"] < sum(s["trans_cubes"] for s in sts) < 1.5 * str1["trans_cubes"]
    assert sts[0]["icp_runs"] == sts[1]["icp_runs"]  # cooperative flow: every refinement is run by both ranks, at the same points of the replicated control flow
    m.close()


def test_dragon_bounds_match_the_oracle_and_add_over_a_source_split(fg, oracle, dragon):
    """kernComputeBounds + the two reductions (registration.cu:27-60, :126-140) on all 437 645 points: against the oracle
    (its LUT filled from the device LUT, which is pinned by brute-force node values first), and additive over a split of
    the source cloud (three contexts, same target, same LUT)."""
    pct, pcs, bounds = dragon["pct"], dragon["pcs"], dragon["bounds"]
    hip = fg.Registration(pct, pcs, bounds, 0.005)
    dx, dy, dz = hip.lut_dims()
    rng = np.random.default_rng(21)
    xyz = np.stack([rng.integers(0, dx, 400), rng.integers(0, dy, 400), rng.integers(0, dz, 400)], 1).astype(np.int32)
    assert np.array_equal(hip.lut_nodes(xyz).view(np.uint32), brute_nodes(pct, bounds, 0.005, xyz).view(np.uint32))
    lut = hip.lut_read()
    assert np.array_equal(lut[xyz[:, 2], xyz[:, 1], xyz[:, 0]].view(np.uint32), hip.lut_nodes(xyz).view(np.uint32))
    orc = oracle.Registration(pct, pcs, bounds, 0.005, build_lut=False)
    assert orc.lut_dims() == (dx, dy, dz)
    orc.lut_set(lut)
    h = len(pcs) // 2 + 12345
    a = fg.Registration(pct, pcs[:h], bounds, 0.005)
    b = fg.Registration(pct, pcs[h:], bounds, 0.005)
    for fix, rn, span in ((True, fg.RotNode(0.25, -0.125, 0.375, 0.125), 0.25), (False, fg.RotNode(-0.1, 0.3, 0.05, 0.0625), 0.0625)):
        tn = tnodes(rng, 32, span)
        lb, ub = hip.compute_sse_error(rn, tn, fix)
        lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
        assert np.allclose(ub, ubo, rtol=1e-6, atol=0) and np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * float(ubo.max()))
        la, ua = a.compute_sse_error(rn, tn, fix)
        lb2, ub2 = b.compute_sse_error(rn, tn, fix)
        assert np.allclose(ua.astype(np.float64) + ub2, ub, rtol=1e-6) and np.allclose(la.astype(np.float64) + lb2, lb, rtol=1e-6, atol=1e-6 * float(ub.max()))
    for r in (hip, a, b):
        r.close()


def test_dragon_scan_equals_brute_force_on_the_whole_cloud(fg, dragon):
    """437 645 x 437 645: the exact box scan against the O(ns * nt) kernels (both HIP; the brute-force kernels are the ones the
    small-size tests compare with the oracle) — SSE bits (registration.cu:14-25, :162-174) and every correspondence
    (icp3d.cu:11-28).  LUT resolution 0.02: the brute-force context builds its LUT by brute force too."""
    pct, pcs, bounds = dragon["pct"], dragon["pcs"], dragon["bounds"]
    scan = fg.Registration(pct, pcs, bounds, 0.02)
    brute = fg.Registration(pct, pcs, bounds, 0.02, flags=fg.FLAG_BRUTE_FORCE_NN)
    assert np.array_equal(scan.lut_read().view(np.uint32), brute.lut_read().view(np.uint32))
    rng = np.random.default_rng(1)
    R = fg.synth.random_rotation(rng, 20.0).astype(f32)
    t = rng.uniform(-0.05, 0.05, 3).astype(f32)
    assert scan.compute_sse_error(R, t).view(np.uint32) == brute.compute_sse_error(R, t).view(np.uint32)
    w = (pcs @ R.T + t).astype(f32)
    *_, idx_a = scan.procrustes(w)
    *_, idx_b = brute.procrustes(w)
    assert np.array_equal(idx_a, idx_b)
    scan.close(); brute.close()


# ------------------------------------------------------------------------------------------------
# (b) 1 M points, 20 % outliers, trimmed
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def million(fg, gpu_required):
    tgt, src, R_gt, t_gt = fg.synth.workload("synthetic1m_outliers", angle_deg=150.0, min_angle_deg=110.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    t_scaled = (float(scale) * (R_gt @ (-off_s.astype(np.float64)) + t_gt + off_t.astype(np.float64))).astype(f32)  # ground truth in the scaled frame
    return dict(tgt=tgt, src=src, R_gt=R_gt, t_gt=t_gt, pct=pct, pcs=pcs, bounds=bounds, k=int(len(src) * 0.8), t_scaled=t_scaled)


def test_trimmed_1m_bounds_equal_partition_of_device_values_and_the_oracle(fg, oracle, million):
    m = million
    k = m["k"]
    hip = fg.Registration(m["pct"], m["pcs"], m["bounds"], 0.005)
    hip.set_inliers(k)
    lut = hip.lut_read()
    orc = oracle.Registration(m["pct"], m["pcs"], m["bounds"], 0.005, build_lut=False)
    orc.lut_set(lut)
    orc.set_inliers(k)
    rng = np.random.default_rng(8)
    cases = [(True, fg.RotNode(0.25, -0.125, 0.375, 0.125), 0.25), (False, fg.RotNode(-0.1, 0.3, 0.05, 0.0625), 0.0625),
             (False, fg.RotNode(0.5, 0.5, -0.5, 0.5), 0.5)]  # the last: most points inside the rotation radius (e = 0)
    for ci, (fix, rn, span) in enumerate(cases):
        tn = tnodes(rng, 6, span, 0.3)
        lb, ub = hip.compute_sse_error(rn, tn, fix)
        lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
        assert np.allclose(ub, ubo, rtol=1e-6, atol=1e-12) and np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * max(float(ubo.max()), 1e-12))
        # the per-point values behind subcube 0, read back from the device: bit-identical to the numpy restatement, and the
        # device's trimmed sums are the sums of their k smallest
        e = hip.point_distances(rn.q.R, rn.span, tn[0], fix)
        if ci < 2:
            assert np.array_equal(e.view(np.uint32), npr.point_distances(lut, m["bounds"], 0.005, m["pcs"], rn.q.R, rn.span, tn[0], fix).view(np.uint32))
        ub_i = (e * e).astype(f32)
        l = (e - f32(SQRT3 * f32(span))).astype(f32)
        lb_i = np.where(l > 0, (l * l).astype(f32), f32(0))
        ub_k = np.partition(ub_i, k - 1)[:k].astype(np.float64).sum()
        lb_k = np.partition(lb_i, k - 1)[:k].astype(np.float64).sum()
        assert float(ub[0]) == pytest.approx(ub_k, rel=1e-6, abs=1e-12) and float(lb[0]) == pytest.approx(lb_k, rel=1e-6, abs=1e-6 * max(ub_k, 1e-12))
    hip.close()


def test_trimmed_1m_single_row_selections_and_pruned_nn_agree(fg, million, monkeypatch):
    """Trimmed SSE and trimmed ICP at 1M points: device-wide selection == one-block selection, and the exact search pruned by
    the LUT brackets (only queries that can be among the k smallest are searched) == the unpruned search, bit for bit."""
    m = million
    rng = np.random.default_rng(5)
    R = (fg.synth.random_rotation(rng, 3.0) @ m["R_gt"]).astype(f32)  # near the ground truth: the regime ICP runs in
    t = (m["t_scaled"] + rng.uniform(-0.02, 0.02, 3)).astype(f32)
    out = {}
    for name, wide, skip in (("default", "1", "1"), ("one_block", "0", "1"), ("unpruned", "1", "0")):
        monkeypatch.setenv("FGOICP_SELECT_WIDE", wide)
        monkeypatch.setenv("FGOICP_TRIM_SKIP", skip)
        reg = fg.Registration(m["pct"], m["pcs"], m["bounds"], 0.005)
        reg.set_inliers(m["k"])
        sse = reg.compute_sse_error(R, t)
        icp = fg.IterativeClosestPoint3D(reg, None, None, 6, 0.0, R, t)
        e, Ri, ti = icp.run()
        out[name] = (sse, e, Ri, ti, icp.iterations)
        reg.close()
    d, o, u = out["default"], out["one_block"], out["unpruned"]
    assert d[0].view(np.uint32) == u[0].view(np.uint32) and d[1].view(np.uint32) == u[1].view(np.uint32)
    assert np.array_equal(d[2], u[2]) and np.array_equal(d[3], u[3]) and d[4] == u[4] == o[4] and d[4] >= 2
    assert float(d[0]) == pytest.approx(float(o[0]), rel=1e-6) and float(d[1]) == pytest.approx(float(o[1]), rel=1e-5)
    assert np.allclose(d[2], o[2], atol=1e-5) and np.allclose(d[3], o[3], atol=1e-5)
    assert float(d[1]) < float(d[0])  # six ICP steps improved the trimmed error


def test_trimmed_1m_full_run_recovers_the_ground_truth(fg, million):
    m = million
    s = fg.FastGoICP(m["tgt"], m["src"], 0.005, 1e-3, schedule=fg.SCHEDULE_ROUND, round_width=0, trim_fraction=0.2)
    R, t = s.run()
    st = s.stats()
    assert ang_deg(R, m["R_gt"]) < 0.15 and np.linalg.norm(t - m["t_gt"]) < 2e-3 * 0.2
    assert st["trans_cubes"] > 10000 and s.registration.sort_fallbacks()[1] == 0
    s.close()


def test_trimmed_1m_on_two_ranks_with_split_scans(fg, million):
    """BASELINE configs[4] sharded: the 1 M-point trimmed pair on two in-process ranks.  Trimmed contexts of this size split the exact
    scans of every refinement over the ranks (device all-gathers of the per-query results): the run must recover the ground
    truth, both ranks must hold the same incumbent bit for bit, device all-gathers must have happened, and the optimum must be the
    one-GPU trimmed run's to 1e-5."""
    m = million
    one = fg.FastGoICP(m["tgt"], m["src"], 0.005, 1e-3, schedule=fg.SCHEDULE_ROUND, round_width=0, trim_fraction=0.2)
    R1, t1 = one.run()
    e1 = one.get_best_error()
    one.close()
    mm = fg.MultiGoICP(m["tgt"], m["src"], 0.005, 1e-3, devices=[0, 0], transport=fg.TRANSPORT_IN_PROCESS, schedule=fg.SCHEDULE_ROUND, round_width=0, trim_fraction=0.2)
    mm.set_record(True)
    R, t = mm.run()
    assert ang_deg(R, m["R_gt"]) < 0.15 and np.linalg.norm(t - m["t_gt"]) < 2e-3 * 0.2
    assert same_result((R, t, mm.get_best_error()), (R1, t1, e1))
    assert mm.get_best_error(1) == mm.get_best_error(0)
    host_ex, dev_gathers = mm.recorded(0)
    assert host_ex > 0 and dev_gathers > 0
    assert mm.replay_rank(0) > 0 and mm.get_best_error(0) == mm.get_best_error(1)
    mm.close()


# ------------------------------------------------------------------------------------------------
# (c) test/bunny.toml shape: nt ~ 0.5 * 35 947, ns ~ 0.1 * 30 379, lut_resolution 0.002
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def toml_shape(fg, gpu_required):
    tgt, src, R_gt, t_gt = fg.synth.workload("bunny_toml", angle_deg=150.0, min_angle_deg=110.0)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    return dict(tgt=tgt, src=src, R_gt=R_gt, t_gt=t_gt, pct=pct, pcs=pcs, bounds=bounds)


def test_bunny_toml_lut_at_resolution_0002(fg, oracle, toml_shape):
    """buildLUTKernel (registration.cu:258-278) and NearestNeighborLUT::search (:320-328) on the ~6e8-node LUT of
    test/bunny.toml: the 8 corner nodes of 1500 random cells == single-node brute force, and lookups inside those cells are
    bit-exact against the CUDA linear-filtering restatement evaluated on those corners."""
    c = toml_shape
    res = 0.002
    hip = fg.Registration(c["pct"], c["pcs"], c["bounds"], res)
    dx, dy, dz = hip.lut_dims()
    assert (dx, dy, dz) == npr.lut_dims(c["bounds"], res) and dx * dy * dz > 4e8
    rng = np.random.default_rng(3)
    ncell = 1500
    cell = np.stack([rng.integers(0, dx - 1, ncell), rng.integers(0, dy - 1, ncell), rng.integers(0, dz - 1, ncell)], 1).astype(np.int32)
    cell[:8] = [[0, 0, 0], [dx - 2, dy - 2, dz - 2], [0, dy - 2, 0], [dx - 2, 0, 0], [0, 0, dz - 2], [dx // 2, dy // 2, dz // 2], [1, 1, 1], [dx - 2, dy - 2, 0]]
    corners = (cell[:, None, :] + np.array([[i, j, k] for k in (0, 1) for j in (0, 1) for i in (0, 1)], np.int32)[None, :, :]).reshape(-1, 3)
    got = hip.lut_nodes(corners)
    assert np.array_equal(got.view(np.uint32), brute_nodes(c["pct"], c["bounds"], res, corners).view(np.uint32))
    # lookups: u - 0.5 = cell + frac on every axis, so the footprint of the lookup is exactly the cell's 8 corners
    frac = rng.uniform(0.05, 0.95, (ncell, 3))
    q = ((cell + 0.5 + frac) * res + c["bounds"][:, 0].astype(np.float64)[None, :]).astype(f32)
    b = np.asarray(c["bounds"], f32)
    scale = f32(1.0) / f32(res)
    w = []
    for a in range(3):
        u = ((q[:, a] + (-b[a, 0])).astype(f32) * scale).astype(f32)
        i0, i1, wa = npr._axis(u, (dx, dy, dz)[a], True)
        assert np.array_equal(i0, cell[:, a]) and np.array_equal(i1, cell[:, a] + 1)
        w.append(wa)
    v = got.reshape(ncell, 2, 2, 2)  # [cell][z][y][x]
    lerp = lambda p, q_, w_: npr.fma(w_, (q_ - p).astype(f32), p)
    c00 = lerp(v[:, 0, 0, 0], v[:, 0, 0, 1], w[0]); c10 = lerp(v[:, 0, 1, 0], v[:, 0, 1, 1], w[0])
    c01 = lerp(v[:, 1, 0, 0], v[:, 1, 0, 1], w[0]); c11 = lerp(v[:, 1, 1, 0], v[:, 1, 1, 1], w[0])
    want = lerp(lerp(c00, c10, w[1]), lerp(c01, c11, w[1]), w[2])
    assert np.array_equal(hip.lut_search(q).view(np.uint32), want.view(np.uint32))
    # and the oracle's own search on queries far outside the grid (clamp addressing needs only border nodes): constant planes
    far = np.array([[1e6, 1e6, 1e6], [-1e6, -1e6, -1e6]], f32)
    corner_nodes = hip.lut_nodes(np.array([[dx - 1, dy - 1, dz - 1], [0, 0, 0]], np.int32))
    assert np.array_equal(hip.lut_search(far).view(np.uint32), corner_nodes.view(np.uint32))
    hip.close()


@pytest.mark.parametrize("mse", [1e-3, 1e-4])
def test_bunny_toml_shape_runs(fg, toml_shape, mse):
    """test/bunny.toml's parameters (lut_resolution 0.002; mse_threshold 1e-3 = the file's, 1e-4 = below the residual, so the
    search has to certify) on the synthetic pair of its size: the CLI's schedule (SERIAL, src/main.cpp:46-51) and expansion
    rounds end in the same optimum, which is the ground truth.  Tolerance: Go-ICP returns an eps-optimal solution, and the
    final refinement ICP stops when an iteration improves the error by less than 0.05 % (fgoicp.cpp:22-23) — two schedules
    that reach the basin from different cubes agree to that band, not to 1e-5 (at 1e-3 the threshold, ns * mse = 3, is six
    times the residual and the search ends with the first good ICP)."""
    c = toml_shape
    res = {}
    for name, sched, K in (("serial", fg.SCHEDULE_SERIAL, 1), ("round", fg.SCHEDULE_ROUND, 0)):
        s = fg.FastGoICP(c["tgt"], c["src"], 0.002, mse, schedule=sched, round_width=K)
        R, t = s.run()
        res[name] = (R, t, float(s.get_best_error()), s.stats())
        assert s.stats()["trans_cubes"] > 0
        s.close()
    (Rs, ts, es, sts), (Rr, tr, er, _) = res["serial"], res["round"]
    print(f"mse {mse}: serial sse {es:.7f} ({sts['trans_cubes']} subcubes), round sse {er:.7f}, |dR| {np.abs(Rs - Rr).max():.2e}")
    assert abs(es - er) <= 2e-3 * er and np.allclose(Rs, Rr, atol=2e-3) and np.allclose(ts, tr, atol=2e-3 * 0.156)
    assert abs(es - er) <= len(c["src"]) * mse  # both are eps-optimal
    assert ang_deg(Rs, c["R_gt"]) < 0.5 and np.linalg.norm(ts - c["t_gt"]) < 1e-3


# ------------------------------------------------------------------------------------------------
# early exit of subcubes the inner branch-and-bound drops anyway (fgoicp_bounds_submit_cut)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sched_name", ["serial", "round"])
def test_early_exit_leaves_the_search_alone_on_the_benchmark_run(fg, gpu_required, sched_name):
    """The benchmark's certify run (bunny shape, mse 5e-5) with the early exit (the default) and with every subcube evaluated in full,
    as the reference does: the same counters, the same incumbent bit for bit — and the kernel skipped a good part of its work items.
    (SERIAL's counters are in turn those of the oracle's literal driver: tests/test_gpu_cli_dist.py, tests/test_host_logic.py.)"""
    tgt, src, R_gt, t_gt = fg.synth.workload("bunny", angle_deg=150.0, min_angle_deg=110.0)
    sched, K = (fg.SCHEDULE_SERIAL, 1) if sched_name == "serial" else (fg.SCHEDULE_ROUND, 0)
    out = {}
    for on in (False, True):
        s = fg.FastGoICP(tgt, src, 0.005, 5e-5, schedule=sched, round_width=K)
        s.set_early_exit(on)
        reg = s.registration
        reg.cut_stats(reset=True)
        R, t = s.run()
        out[on] = (R, t, float(s.get_best_error()), s.stats(), reg.cut_stats())
        s.close()
    (R0, t0, e0, st0, c0), (R1, t1, e1, st1, c1) = out[False], out[True]
    assert np.array_equal(R0, R1) and np.array_equal(t0, t1) and e0 == e1
    for k in ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb", "rounds"):
        assert st0[k] == st1[k], k
    assert c0 == (0, 0)
    offered, skipped = c1
    print(f"{sched_name}: {st1['trans_cubes']} subcubes, {offered} work items, {skipped} not evaluated ({skipped / offered:.3f})")
    assert offered > 0 and skipped > 0.25 * offered
    assert ang_deg(R1, R_gt) < 0.5


def test_every_window_with_thresholds_keeps_the_contract(fg, gpu_required, monkeypatch, capfd):
    """Development build, FGOICP_CUT_VERIFY=1: every window of a certify run that carried thresholds is evaluated once more without them, and
    every row is checked — at or above its threshold T in the exact evaluation: {T, T} was reported; below: the exact bits (ctx.hip
    tick_wait_window).  (profiles/r04_early_exit_contract_verified.txt: the four benchmark runs, 6.8 M rows, 0 violations.)"""
    import re
    monkeypatch.setenv("FGOICP_CUT_VERIFY", "1")
    tgt, src, R_gt, t_gt = fg.synth.workload("small", angle_deg=150.0, min_angle_deg=110.0)
    for sched, K in ((fg.SCHEDULE_SERIAL, 1), (fg.SCHEDULE_ROUND, 0)):
        s = fg.FastGoICP(tgt, src, 0.02, 2e-5, schedule=sched, round_width=K)
        s.run()
        assert s.stats()["trans_cubes"] > 1000
        s.close()
        err = capfd.readouterr().err
        m = re.search(r"cut verify\] (\d+) windows .*: (\d+) rows, (\d+) of them at or above their threshold, (\d+) violations", err)
        assert m, err[-400:]
        windows, rows, above, bad = map(int, m.groups())
        assert windows > 0 and rows > 1000 and above > 0 and bad == 0


# ------------------------------------------------------------------------------------------------
# the tick sort's permutation check
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("xcd", ["1", "0"])
def test_tick_sort_is_checked_on_the_device_and_falls_back(fg, tiny_case, gpu_required, monkeypatch, xcd):
    """Every sorted tick is verified on the device to be a permutation of its work items (both histogram flavours); a spoilt
    sort (test hook: one slot left unwritten) is detected, the tick is repeated with device-scope atomics, results equal."""
    c = tiny_case
    rng = np.random.default_rng(12)
    nodes = [fg.RotNode(*rng.uniform(-0.4, 0.4, 3), 0.125) for _ in range(6)]
    groups = [tnodes(rng, 300, 0.25, 0.6) for _ in nodes]
    fixes = [bool(i % 2) for i in range(len(nodes))]
    args = ([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
    monkeypatch.setenv("FGOICP_SORT_XCD", xcd)
    ref = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    want = ref.compute_bounds_multi(*args)
    ticks, fb = ref.sort_fallbacks()
    assert ticks >= 1 and fb == 0
    ref.close()
    reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    reg.test_sort_fault(2)  # a test hook of the ABI: the second sorted tick from now gets a spoiled slot
    for rep in range(3):
        got = reg.compute_bounds_multi(*args)
        for (lb, ub), (lbw, ubw) in zip(got, want):
            assert np.array_equal(lb, lbw) and np.array_equal(ub, ubw)
    ticks, fb = reg.sort_fallbacks()
    assert ticks >= 3 and fb == 1
    reg.close()
    # the same with thresholds (fgoicp_bounds_submit_cut: two tiers, running sums, early exits): the repeated window starts from clean sums
    reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    cut = np.array([np.median(lbw) for lbw, _ in want], np.float32)
    reg.test_sort_fault(2)
    for rep in range(3):
        got = reg.compute_bounds_cut(*args, cut)
        for g, ((lb, ub), (lbw, ubw)) in enumerate(zip(got, want)):
            below = lbw < cut[g]
            assert np.array_equal(lb[below], lbw[below]) and np.array_equal(ub[below], ubw[below]) and np.all(lb[~below] == cut[g]) and np.all(ub[~below] == cut[g])
    ticks, fb = reg.sort_fallbacks()
    assert ticks >= 3 and fb == 1
    reg.close()


# ------------------------------------------------------------------------------------------------
# configs[0] on the REAL clouds: the inputs of test/bunny.toml as the CLI's loader produces them (tests/golden/bunny_toml_clouds.npz)
# ------------------------------------------------------------------------------------------------
def test_bunny_toml_on_the_stanford_bunny_clouds(fg, oracle, gpu_required):
    """The reference's example run (test/bunny.toml: model_bunny.txt at 0.5, data_bunny.txt at 0.1, lut_resolution 0.002,
    mse_threshold 1e-3) through FastGoICP::run() in the CLI's schedule.  The oracle cannot build a 6e8-node LUT, but its EXACT
    operators need none: the residual of the returned (R, t) is re-evaluated by the oracle's brute-force SSE on the oracle's own
    pre-processing (registration.cu:62-86, fgoicp.cpp:176-287), and the oracle's ICP continued from there stays in the same basin (the
    refinement stops when an iteration gains less than 0.05 %, fgoicp.cpp:22-23, so a continued ICP may still creep by a fraction of a
    percent).  Expansion rounds end in the same optimum, to the band two eps-optimal runs agree to (see test_bunny_toml_shape_runs)."""
    import os
    F = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bunny_toml_clouds.npz"))
    tgt, src = F["tgt"], F["src"]
    res = {}
    for name, sched, K in (("serial", fg.SCHEDULE_SERIAL, 1), ("round", fg.SCHEDULE_ROUND, 0)):
        s = fg.FastGoICP(tgt, src, float(F["lut_resolution"]), float(F["mse_threshold"]), schedule=sched, round_width=K)
        if name == "serial":
            dims = s.registration.lut_dims()
            assert all(600 < d < 1000 for d in dims) and dims[0] * dims[1] * dims[2] > 4e8  # SURVEY: 923 x 906 x 711 on the full clouds
        R, t = s.run()
        R_s, t_s = s.get_best_transform()  # scaled frame
        res[name] = (R, t, float(s.get_best_error()), s.stats(), R_s, t_s)
        s.close()
    (Rs, ts, es, st, R_scaled, t_scaled), (Rr, tr, er, *_) = res["serial"], res["round"]
    print(f"bunny.toml: sse {es:.6f} ({st['trans_cubes']} subcubes, {st['rot_cubes']} rotation cubes, {st['icp_runs']} ICP runs); round sse {er:.6f}")
    assert st["trans_cubes"] > 1000 and st["icp_runs"] >= 2
    assert abs(es - er) <= 2e-3 * es and np.allclose(Rs, Rr, atol=2e-3)
    # the oracle on the same inputs: its pre-processing, its exact SSE of the returned transform
    o = oracle.FastGoICP(tgt, src, 0.5, 1e-3)  # the LUT resolution only sizes a LUT this check never reads
    pp = o.preproc()
    reg = oracle.Registration(pp["pct"], pp["pcs"], pp["bounds"], 0.5, build_lut=False)
    assert float(reg.compute_sse_error(R_scaled, t_scaled)) == pytest.approx(es, rel=1e-5)
    sse_o, R_o, t_o, it = reg.icp(R_scaled, t_scaled, 100, 0.0005)
    assert es * (1 - 0.02) <= float(sse_o) <= es * (1 + 1e-5) and np.allclose(R_o, R_scaled, atol=1e-2)
    # restore_translation (fgoicp.hpp:87-90) with the oracle's offsets and scale
    t_rest = t_scaled.astype(np.float64) / float(pp["scale"]) + R_scaled.astype(np.float64) @ pp["offset_pcs"] - pp["offset_pct"]
    assert np.allclose(t_rest, ts, atol=1e-5)
