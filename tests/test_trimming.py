"""EXTENSION — trimmed Go-ICP.  The reference parses `params.trim` and ignores it, so there is
no reference behaviour; the definition follows Yang et al.'s Go-ICP (sum of the inlierNum smallest per-point
terms).  CPU: the oracle against numpy and against the product's host driver; GPU: HIP against the oracle."""
import os

import numpy as np
import pytest

from oracle import np_restatement as npr
from tests import host_harness as hh

f32 = np.float32
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goicp_golden.npz"))


def outlier_pair(fg, nt=700, ns=500, frac=0.2, seed=9, angle=(100.0, 120.0)):
    """Source = exact copies of target points under a known motion, `frac` of them replaced by uniform outliers."""
    tgt, _, _, _ = fg.synth.make_pair(nt, 8, (0.156, 0.152, 0.118), seed=seed)
    rng = np.random.default_rng(seed + 1)
    R_gt = fg.synth.random_rotation(rng, angle[1], angle[0])
    t_gt = rng.uniform(-0.02, 0.02, 3)
    src = ((tgt[:ns].astype(np.float64) - t_gt) @ R_gt).astype(f32)  # R_gt @ src + t_gt == tgt[:ns]
    n_out = int(ns * frac)
    idx = rng.choice(ns, n_out, replace=False)
    src[idx] = rng.uniform(-0.12, 0.12, (n_out, 3)).astype(f32)
    return tgt, src, R_gt, t_gt


@pytest.fixture(scope="module")
def small_case(oracle):
    rng = np.random.default_rng(3)
    tgt = rng.uniform(-0.5, 0.45, size=(60, 3)).astype(f32)
    src = rng.uniform(-0.4, 0.4, size=(97, 3)).astype(f32)
    bounds = np.array([[tgt[:, k].min(), tgt[:, k].max()] for k in range(3)], f32)
    return tgt, src, bounds, 0.07


@pytest.mark.parametrize("k", [1, 40, 96])
def test_oracle_trimmed_bounds_and_sse_match_numpy(oracle, small_case, k):
    tgt, src, bounds, res = small_case
    reg = oracle.Registration(tgt, src, bounds, res)
    R, _, _ = oracle.rotation(0.2, -0.3, 0.1)
    rng = np.random.default_rng(2)
    tn = np.concatenate([rng.uniform(-0.3, 0.3, (6, 3)), rng.choice([0.5, 0.25, 0.0625], (6, 1))], 1).astype(f32)
    full = reg.compute_bounds(R, 0.25, tn, False)
    reg.set_inliers(k)
    lb, ub = reg.compute_bounds(R, 0.25, tn, False)
    lbn, ubn = npr.bounds(reg.lut_get(), bounds, res, src, R, 0.25, tn, False, inliers=k)
    assert np.allclose(ub, ubn, rtol=2e-6, atol=1e-9) and np.allclose(lb, lbn, rtol=2e-6, atol=1e-7)
    assert np.all(ub <= full[1] * (1 + 1e-6)) and np.all(lb <= full[0] * (1 + 1e-6) + 1e-9)
    t = np.array([0.05, -0.02, 0.01], f32)
    rp = npr.rot_apply(R, src) + t
    d2 = np.min(npr.dist_sq(rp[:, None, :].astype(f32), tgt[None, :, :]), axis=1)
    assert float(reg.compute_sse_error(R, t)) == pytest.approx(float(np.sort(d2)[:k].astype(np.float64).sum()), rel=2e-6)
    reg.set_inliers(0)
    assert np.array_equal(reg.compute_bounds(R, 0.25, tn, False)[1], full[1])  # k = 0 restores the reference behaviour


def test_oracle_trimmed_procrustes_uses_the_k_closest(oracle):
    rng = np.random.default_rng(5)
    tgt = rng.uniform(-1, 1, (300, 3)).astype(f32)
    work = tgt[:100].copy()
    work[:20] += rng.uniform(0.3, 0.5, (20, 3)).astype(f32)  # 20 gross outliers
    reg = oracle.Registration(tgt, work, np.array([[-1, 1]] * 3, f32), 0.5, build_lut=False)
    R_full, t_full, *_ = reg.procrustes(work)
    reg.set_inliers(80)
    R, t, cen, ABt, idx = reg.procrustes(work)
    assert np.allclose(R, np.eye(3), atol=1e-6) and np.allclose(t, 0, atol=1e-6)       # inliers are exact copies
    assert not np.allclose(t_full, 0, atol=1e-3)                                         # the untrimmed step is dragged away
    assert np.allclose(cen[:3], work[20:].mean(0), atol=1e-6)


def test_trimmed_search_rejects_outliers(oracle, fg):
    """20 % uniform outliers: the plain optimum is biased, the trimmed run finds the ground truth."""
    tgt, src, R_gt, t_gt = outlier_pair(fg, nt=600, ns=400, frac=0.2, seed=9, angle=(20.0, 30.0))
    trim = oracle.FastGoICP(tgt, src, 0.05, 1e-3, trim_fraction=0.25).run()
    plain = oracle.FastGoICP(tgt, src, 0.05, 1e-3).run()
    ang = lambda R: np.degrees(np.arccos(np.clip((np.trace(R.astype(np.float64).T @ R_gt) - 1) / 2, -1, 1)))
    assert ang(trim["R"]) < 0.05 and np.linalg.norm(trim["t"] - t_gt) < 1e-4 and float(trim["best_sse"]) < 1e-6
    assert ang(plain["R"]) > 10 * ang(trim["R"]) and float(plain["best_sse"]) > 1.0


def test_product_driver_with_trimming_follows_the_oracle_driver(oracle, fg):
    tgt, src, R_gt, t_gt = outlier_pair(fg, nt=500, ns=300, frac=0.2, seed=4, angle=(100.0, 130.0))
    o = oracle.FastGoICP(tgt, src, 0.05, 1e-3, trim_fraction=0.25).run()
    h = hh.HostDriver(tgt, src, 0.05, 1e-3, schedule=0, trim_fraction=0.25).run()
    assert [h["stats"][k] for k in ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb")] == \
           [o["stats"][k] for k in ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb")]
    assert np.array_equal(h["R"], o["R"]) and np.array_equal(h["t"], o["t"]) and h["best_sse"] == o["best_sse"]
    r = hh.HostDriver(tgt, src, 0.05, 1e-3, schedule=1, round_width=4, trim_fraction=0.25).run()
    assert float(r["best_sse"]) == pytest.approx(float(o["best_sse"]), rel=1e-5, abs=1e-6) and np.allclose(r["R"], o["R"], atol=1e-5)


# ------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("k_frac", [0.8, 0.5, 0.013])
def test_hip_trimmed_operators_match_oracle(fg, oracle, tiny_case, gpu_required, k_frac):
    c = tiny_case
    k = max(1, int(len(c["pcs"]) * k_frac))
    hip = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    orc = oracle.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    hip.set_inliers(k); orc.set_inliers(k)
    rng = np.random.default_rng(17)
    rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)
    tn = np.concatenate([rng.uniform(-0.6, 0.6, (45, 3)), rng.choice([0.5, 0.25, 0.0625], (45, 1))], 1).astype(f32)
    for fix in (True, False):
        lb, ub = hip.compute_sse_error(rn, tn, fix)
        lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
        assert np.allclose(ub, ubo, rtol=1e-6, atol=1e-12)
        assert np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * max(float(ubo.max()), 1e-12))
    R = fg.synth.random_rotation(rng, 30.0).astype(f32)
    t = rng.uniform(-0.1, 0.1, 3).astype(f32)
    assert float(hip.compute_sse_error(R, t)) == pytest.approx(float(orc.compute_sse_error(R, t)), rel=1e-6)
    w = (c["pcs"] @ R.T + t).astype(f32)
    Rh, th, cen, ABt, idx = hip.procrustes(w)
    Ro, to, ceno, ABto, idxo = orc.procrustes(w)
    # points that provably cannot be among the k closest get no correspondence (index 0x7fffffff, nn_prep_kernel); every other
    # index is the reference's, and the points left out really lie beyond the k-th smallest correspondence distance
    found = idx != 0x7fffffff
    d2o = ((w - c["pct"][idxo]).astype(np.float64) ** 2).sum(1)
    assert np.array_equal(idx[found], idxo[found]) and found.sum() >= k and (found.all() or d2o[~found].min() > np.sort(d2o)[k - 1])
    assert np.allclose(cen, ceno, rtol=1e-6, atol=1e-7)
    assert np.allclose(ABt, ABto, rtol=1e-5, atol=1e-5) and np.allclose(Rh, Ro, atol=2e-6) and np.allclose(th, to, atol=2e-6)
    sse, Ri, ti = fg.IterativeClosestPoint3D(hip, None, None, 100, 0.005, R, t).run()
    sse_o, Ri_o, ti_o, it_o = orc.icp(R, t, 100, 0.005)
    assert float(sse) == pytest.approx(float(sse_o), rel=1e-5) and np.allclose(Ri, Ri_o, atol=1e-5)
    hip.set_inliers(0)  # back to the reference behaviour
    lb, ub = hip.compute_sse_error(rn, tn, False)
    orc.set_inliers(0)
    assert np.allclose(ub, orc.compute_bounds(rn.q.R, rn.span, tn, False)[1], rtol=1e-6)
    hip.close()


@pytest.mark.gpu
def test_hip_trimmed_ties_at_the_cut_take_the_lowest_index(fg, oracle, gpu_required):
    rng = np.random.default_rng(2)
    tgt = rng.uniform(-0.8, 0.8, (50, 3)).astype(f32)
    off = np.array([0.0625, 0.0, 0.0], f32)
    work = np.concatenate([tgt[:10], tgt[10:30] + off, tgt[30:40] + 3 * off]).astype(f32)  # 20 points tie exactly at the cut
    bounds = np.array([[-1, 1]] * 3, f32)
    hip = fg.Registration(tgt, work, bounds, 0.25)
    orc = oracle.Registration(tgt, work, bounds, 0.25, build_lut=False)
    for k in (15, 22, 29):
        hip.set_inliers(k); orc.set_inliers(k)
        Rh, th, cen, *_ = hip.procrustes(work)
        Ro, to, ceno, *_ = orc.procrustes(work)
        assert np.allclose(cen, ceno, rtol=1e-6, atol=1e-7) and np.allclose(Rh, Ro, atol=2e-6) and np.allclose(th, to, atol=2e-6)
    hip.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sched,K", [(0, 1), (1, 4)])
def test_hip_trimmed_full_run_matches_oracle(fg, oracle, gpu_required, sched, K):
    tgt, src, R_gt, t_gt = outlier_pair(fg, nt=500, ns=300, frac=0.2, seed=4, angle=(100.0, 130.0))
    o = oracle.FastGoICP(tgt, src, 0.05, 1e-3, trim_fraction=0.25).run()
    s = fg.FastGoICP(tgt, src, 0.05, 1e-3, schedule=sched, round_width=K, trim_fraction=0.25)
    R, t = s.run()
    assert float(s.get_best_error()) == pytest.approx(float(o["best_sse"]), rel=1e-5, abs=1e-6)
    assert np.allclose(R, o["R"], atol=1e-5) and np.allclose(t, o["t"], atol=1e-5)
    if sched == 0:
        st = s.stats()
        assert [st[k] for k in ("trans_cubes", "rot_cubes", "icp_runs", "icp_iters")] == [o["stats"][k] for k in ("trans_cubes", "rot_cubes", "icp_runs", "icp_iters")]
    s.close()


@pytest.mark.gpu
def test_hip_device_wide_selection_equals_the_one_block_selection(fg, gpu_required, monkeypatch):
    """Trimmed SSE and the ICP inlier cut of a long row (ns >= 32768) are selected by the whole device (nine small launches
    instead of one block walking 40k..1M values four times): same k-th value, same trimmed sum up to the fp64 summation
    order, same ICP result."""
    tgt, src, R_gt, t_gt = fg.synth.workload("bunny", angle_deg=20.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    rng = np.random.default_rng(5)
    pcs = pcs.copy()
    bad = rng.choice(len(pcs), len(pcs) // 5, replace=False)
    pcs[bad] = rng.uniform(-1.2, 1.2, (len(bad), 3)).astype(f32)  # 20 % outliers
    k = int(len(pcs) * 0.8)
    R = fg.synth.random_rotation(rng, 10.0).astype(f32)
    t = rng.uniform(-0.02, 0.02, 3).astype(f32)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FGOICP_SELECT_WIDE", mode)
        reg = fg.Registration(pct, pcs, bounds, 0.02)
        reg.set_inliers(k)
        sse = float(reg.compute_sse_error(R, t))
        icp = fg.IterativeClosestPoint3D(reg, None, None, 20, 0.005, R, t)
        out[mode] = (sse, *icp.run(), icp.iterations)
        reg.set_inliers(0)
        full = float(reg.compute_sse_error(R, t))
        assert sse < 0.5 * full  # the outliers carried most of the untrimmed error
        reg.close()
    (s1, e1, R1, t1, it1), (s0, e0, R0, t0, it0) = out["1"], out["0"]
    assert s1 == pytest.approx(s0, rel=1e-6) and it1 == it0
    assert float(e1) == pytest.approx(float(e0), rel=1e-5) and np.allclose(R1, R0, atol=1e-5) and np.allclose(t1, t0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("ns", [301, 1202, 2051])
def test_hip_trimmed_rows_with_sizes_that_are_not_multiples_of_four(fg, oracle, gpu_required, ns):
    """Rows of the per-point distances start 16-byte aligned (row stride = ns rounded up to 4) and are read as float4 plus a tail:
    sizes with ns % 4 = 1, 2, 3, including one above a whole 1024-thread sweep, against the oracle."""
    tgt, src, _, _ = fg.synth.make_pair(1500, ns, (0.156, 0.152, 0.118), seed=ns)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    hip = fg.Registration(pct, pcs, bounds, 0.05)
    orc = oracle.Registration(pct, pcs, bounds, 0.05)
    rng = np.random.default_rng(ns)
    rn = fg.RotNode(0.2, -0.15, 0.3, 0.125)
    tn = np.concatenate([rng.uniform(-0.5, 0.5, (40, 3)), rng.choice([0.5, 0.125, 0.0625], (40, 1))], 1).astype(f32)
    for k in (ns - 1, int(0.7 * ns), 5):
        hip.set_inliers(k); orc.set_inliers(k)
        for fix in (True, False):
            lb, ub = hip.compute_sse_error(rn, tn, fix)
            lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
            assert np.allclose(ub, ubo, rtol=1e-6, atol=1e-12) and np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * max(float(ubo.max()), 1e-12))
        e = hip.point_distances(rn.q.R, rn.span, tn[3], False)
        assert e.shape == (ns,) and np.array_equal(e.view(np.uint32), npr.point_distances(hip.lut_read(), bounds, 0.05, pcs, rn.q.R, rn.span, tn[3], False).view(np.uint32))
    hip.close()


@pytest.mark.gpu
def test_hip_trimmed_bounds_across_several_windows(fg, oracle, tiny_case, gpu_required, monkeypatch):
    """A trimmed submission larger than one window of per-point rows (forced: 64 subcubes per window) is split, every window runs its
    own selection, the results are those of the oracle — including groups cut in the middle and twin hints that straddle a cut."""
    monkeypatch.setenv("FGOICP_MAX_SUBCUBES", "64")
    c = tiny_case
    hip = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    orc = oracle.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    k = int(len(c["pcs"]) * 0.75)
    hip.set_inliers(k); orc.set_inliers(k)
    rng = np.random.default_rng(23)
    nodes = [fg.RotNode(*rng.uniform(-0.4, 0.4, 3), 0.125) for _ in range(4)]
    groups = [np.concatenate([rng.uniform(-0.6, 0.6, (n, 3)), rng.choice([0.5, 0.25, 0.0625], (n, 1))], 1).astype(f32) for n in (90, 7, 130, 40)]
    fixes = [True, False, False, True]
    got = hip.compute_bounds_multi([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
    for n, g, f, (lb, ub) in zip(nodes, groups, fixes, got):
        lbo, ubo = orc.compute_bounds(n.q.R, n.span, g, f)
        assert np.allclose(ub, ubo, rtol=1e-6, atol=1e-12) and np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * max(float(ubo.max()), 1e-12))
    hip.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sample", ["5", "0"])
@pytest.mark.parametrize("shape", ["cluster", "identical", "mostly_zero"])
def test_hip_trimmed_selection_paths(fg, oracle, gpu_required, shape, sample, monkeypatch):
    """The rare paths of the per-row selection against the oracle: a histogram bin with more members than the gather buffer holds
    (20 000 source points in a 1e-5 cluster: distinct values within one 1/512-octave bin -> refinement passes), all values equal
    (identical points: the cut's range shrinks to one bit pattern, no gather), and rows whose k smallest are all zero.  Under the
    one-pass selection (sample = 5) the first two are rows whose bracket overflows the member segments: the in-kernel fallback."""
    monkeypatch.setenv("FGOICP_TRIM_SAMPLE", sample)
    rng = np.random.default_rng(31)
    tgt = rng.uniform(-0.8, 0.8, (400, 3)).astype(f32)
    bounds = np.array([[tgt[:, a].min(), tgt[:, a].max()] for a in range(3)], f32)
    n = 20000
    if shape == "identical":
        src = np.tile(np.array([[0.31, -0.22, 0.17]], f32), (n, 1))
    else:
        src = (np.array([[0.31, -0.22, 0.17]]) + rng.normal(scale=1e-5 if shape == "cluster" else 0.2, size=(n, 3))).astype(f32)
    hip = fg.Registration(tgt, src, bounds, 0.05)
    orc = oracle.Registration(tgt, src, bounds, 0.05)
    span_r = 0.5 if shape == "mostly_zero" else 0.0625
    rn = fg.RotNode(0.1, -0.2, 0.15, span_r)
    tn = np.concatenate([rng.uniform(-0.3, 0.3, (12, 3)), rng.choice([0.25, 0.0625], (12, 1))], 1).astype(f32)
    for k in (n // 2, n - 3, 17):
        hip.set_inliers(k); orc.set_inliers(k)
        for fix in (True, False):
            lb, ub = hip.compute_sse_error(rn, tn, fix)
            lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
            assert np.allclose(ub, ubo, rtol=1e-6, atol=1e-12) and np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * max(float(ubo.max()), 1e-12)), (shape, k, fix)
    if shape == "mostly_zero":
        assert (hip.point_distances(rn.q.R, rn.span, tn[0], False) == 0).mean() > 0.3
    rows, fallbacks, members = hip.trim_stats()
    if sample == "0":
        assert rows == 0  # the two-pass kernel keeps no statistics
    else:
        assert rows >= 3 * 2 * len(tn) and (fallbacks > 0 if shape in ("cluster", "identical") else True)  # + the row point_distances submitted
    hip.close()


@pytest.mark.gpu
def test_hip_one_pass_selection_equals_the_two_pass_selection(fg, oracle, gpu_required, monkeypatch):
    """The sampled one-pass selection (trim_rows_sampled_kernel: bracket from the row's 1/32 sample, exact check, LDS radix select)
    against the two-pass kernel and the oracle on a surface cloud with 20 % outliers: same bounds to 1e-6 whatever the bracket
    does — default margin (few or no fallbacks), a margin of zero standard deviations (the bracket misses often: fallbacks), a
    huge margin (everything is a member: segment overflow -> fallback) and a denser sample."""
    tgt, src, R_gt, t_gt = fg.synth.make_pair(4000, 30001, (0.156, 0.152, 0.118), seed=12)
    rng = np.random.default_rng(13)
    bad = rng.choice(len(src), len(src) // 5, replace=False)
    src = src.copy()
    src[bad] = rng.uniform(-0.12, 0.12, (len(bad), 3)).astype(f32)  # 20 % uniform outliers
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    orc = oracle.Registration(pct, pcs, bounds, 0.05)
    rng = np.random.default_rng(8)
    rn = fg.RotNode(0.2, -0.1, 0.3, 0.0625)
    tn = np.concatenate([rng.uniform(-0.4, 0.4, (24, 3)), rng.choice([0.5, 0.125, 0.03125], (24, 1))], 1).astype(f32)
    ks = (int(0.8 * len(pcs)), int(0.3 * len(pcs)), len(pcs) - 1, 2)
    want = {}
    for k in ks:
        orc.set_inliers(k)
        for fix in (True, False):
            want[k, fix] = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
    got = {}
    for mode, env in {"two_pass": {"FGOICP_TRIM_SAMPLE": "0"}, "default": {}, "margin0": {"FGOICP_TRIM_MARGIN": "0"},
                      "wide": {"FGOICP_TRIM_MARGIN": "1000"}, "dense": {"FGOICP_TRIM_SAMPLE": "3"}}.items():
        for name in ("FGOICP_TRIM_SAMPLE", "FGOICP_TRIM_MARGIN"):
            monkeypatch.delenv(name, raising=False)
        for name, val in env.items():
            monkeypatch.setenv(name, val)
        hip = fg.Registration(pct, pcs, bounds, 0.05)
        for k in ks:
            hip.set_inliers(k)
            for fix in (True, False):
                lb, ub = hip.compute_sse_error(rn, tn, fix)
                lbo, ubo = want[k, fix]
                assert np.allclose(ub, ubo, rtol=1e-6, atol=1e-12) and np.allclose(lb, lbo, rtol=1e-6, atol=1e-6 * max(float(ubo.max()), 1e-12)), (mode, k, fix)
                got[mode, k, fix] = (lb.copy(), ub.copy())
        st = hip.trim_stats()
        hip.close()
        if mode == "two_pass":
            assert st[0] == 0
        else:
            assert st[0] == len(ks) * 2 * len(tn)
            if mode == "default":
                assert st[1] <= st[0] // 10, st   # the bracket holds on a surface cloud
            if mode in ("margin0", "wide"):
                assert st[1] > 0, (mode, st)      # ... and these two exercise the fallback
    for mode in ("default", "margin0", "wide", "dense"):
        for k in ks:
            for fix in (True, False):
                assert np.allclose(got[mode, k, fix][1], got["two_pass", k, fix][1], rtol=3e-7, atol=0) and \
                       np.allclose(got[mode, k, fix][0], got["two_pass", k, fix][0], rtol=3e-7, atol=1e-7 * float(got["two_pass", k, fix][1].max()))


@pytest.mark.gpu
def test_hip_trimmed_search_with_cropped_target_bounds(fg, oracle, gpu_required):
    """The reference's Registration takes ANY target_bounds (they only place the LUT, registration.hpp:68).  With bounds cropped to
    a corner of the target, the trimmed search's "distance to the target's box" prune must still use the box of the POINTS
    (taken from the caller's bounds it silently dropped queries whose nearest neighbour lay outside the crop)."""
    tgt, src, R_gt, t_gt = outlier_pair(fg, nt=900, ns=600, frac=0.25, seed=21)
    lo, hi = tgt.min(0), tgt.max(0)
    crop = np.stack([lo, lo + 0.35 * (hi - lo)], 1).astype(f32)  # a third of the extent per axis: most targets lie outside
    hip = fg.Registration(tgt, src, crop, 0.01)
    orc = oracle.Registration(tgt, src, crop, 0.01)
    k = int(0.7 * len(src))
    hip.set_inliers(k); orc.set_inliers(k)
    rng = np.random.default_rng(2)
    for _ in range(4):
        R = fg.synth.random_rotation(rng, 60.0).astype(f32)
        t = rng.uniform(-0.05, 0.05, 3).astype(f32)
        a, b = float(hip.compute_sse_error(R, t)), float(orc.compute_sse_error(R, t))
        assert a == pytest.approx(b, rel=1e-6, abs=1e-12)
    sse, Ri, ti = fg.IterativeClosestPoint3D(hip, None, None, 30, 0.005, R_gt.astype(f32), t_gt.astype(f32)).run()
    sse_o, Ro, to, it = orc.icp(R_gt.astype(f32), t_gt.astype(f32), 30, 0.005)
    assert float(sse) == pytest.approx(float(sse_o), rel=1e-5, abs=1e-10) and np.allclose(Ri, Ro, atol=1e-5)
    hip.close()
