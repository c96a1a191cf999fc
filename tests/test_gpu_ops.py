"""GPU parity tests of the operator level (through the C ABI) against the CPU oracle.

Bars: LUT nodes, LUT lookups, nearest-neighbour indices — bit-exact.  Sums over points
(bounds, SSE, centroids, covariance) — the per-point fp32 values are identical, both sides
accumulate in fp64 and round once, in different orders: tolerance 1e-6 relative
(north_star allows 1e-5 on the final result)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 1e-6


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))


@pytest.fixture(scope="module")
def regs(fg, oracle, tiny_case, gpu_required):
    c = tiny_case
    hip = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    orc = oracle.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    yield hip, orc
    hip.close()


def _tnodes(rng, B, span):
    t = rng.uniform(-0.6, 0.6, size=(B, 3)).astype(np.float32)
    return np.concatenate([t, np.full((B, 1), span, np.float32)], axis=1)


def test_lut_nodes_bit_exact(regs):
    hip, orc = regs
    assert hip.lut_dims() == orc.lut_dims()
    a, b = hip.lut_read(), orc.lut_get()
    assert a.shape == b.shape
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_lut_search_bit_exact(regs, tiny_case):
    hip, orc = regs
    rng = np.random.default_rng(5)
    lo, hi = tiny_case["bounds"][:, 0], tiny_case["bounds"][:, 1]
    inside = rng.uniform(lo, hi, size=(20000, 3))
    outside = rng.uniform(lo - 1.5, hi + 1.5, size=(20000, 3))  # exercises clamp addressing
    edge = np.concatenate([lo[None, :] + rng.uniform(-0.06, 0.06, (2000, 3)), hi[None, :] + rng.uniform(-0.06, 0.06, (2000, 3))])
    far = np.array([[1e6, -1e6, 0.0], [-3e38, 3e38, 1.0]], np.float32)
    q = np.concatenate([inside, outside, edge, far]).astype(np.float32)
    a, b = hip.lut_search(q), orc.lut_search(q)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_lut_search_no_quant_flag(fg, oracle, tiny_case, gpu_required):
    c = tiny_case
    hip = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"], flags=fg.FLAG_NO_WEIGHT_QUANT)
    orc = oracle.Registration(c["pct"], c["pcs"], c["bounds"], c["res"], quantize=False)
    q = np.random.default_rng(6).uniform(-1.2, 1.2, size=(5000, 3)).astype(np.float32)
    assert np.array_equal(hip.lut_search(q).view(np.uint32), orc.lut_search(q).view(np.uint32))
    hip.close()


@pytest.mark.parametrize("fix_rot", [True, False])
@pytest.mark.parametrize("B,span", [(32, 0.25), (1, 1.0), (7, 0.0625), (33, 0.5), (100, 0.125)])
def test_bounds_batch_parity(regs, fg, fix_rot, B, span):
    hip, orc = regs
    rng = np.random.default_rng(100 + B)
    rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)
    tn = _tnodes(rng, B, span)
    lb, ub = hip.compute_sse_error(rn, tn, fix_rot)
    lb_o, ub_o = orc.compute_bounds(rn.q.R, rn.span, tn, fix_rot)
    assert lb.shape == (B,) and ub.shape == (B,)
    assert rel(ub, ub_o) <= REL, (ub, ub_o)
    # lower bounds can be exactly 0 for large spans: compare absolutely against the ub scale
    assert np.max(np.abs(lb.astype(np.float64) - lb_o)) <= REL * np.max(ub_o)
    assert np.all(lb <= ub * (1 + 1e-6))


def test_bounds_multi_equals_batches(regs, fg):
    hip, _ = regs
    rng = np.random.default_rng(9)
    nodes = [fg.RotNode(0.5, 0.5, -0.5, 0.5), fg.RotNode(0.0, 0.125, 0.0, 0.25), fg.RotNode(-0.25, 0.25, 0.25, 0.0625)]
    groups = [_tnodes(rng, 32, 0.5), _tnodes(rng, 5, 0.25), _tnodes(rng, 40, 0.125)]
    fixes = [True, False, False]
    multi = hip.compute_bounds_multi([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
    for n, g, f, (lb, ub) in zip(nodes, groups, fixes, multi):
        lb1, ub1 = hip.compute_sse_error(n, g, f)
        assert np.array_equal(lb, lb1) and np.array_equal(ub, ub1)  # same launches → bit-identical


def test_bounds_run_to_run_deterministic(regs, fg):
    hip, _ = regs
    tn = _tnodes(np.random.default_rng(3), 32, 0.25)
    rn = fg.RotNode(0.1, 0.2, 0.3, 0.25)
    a = hip.compute_sse_error(rn, tn, False)
    b = hip.compute_sse_error(rn, tn, False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_bounds_empty_batch(regs, fg):
    hip, _ = regs
    lb, ub = hip.compute_sse_error(fg.RotNode(0, 0, 0, 1.0), np.zeros((0, 4), np.float32), True)
    assert lb.size == 0 and ub.size == 0


def test_morton_order_is_transparent(fg, tiny_case, regs, gpu_required):
    c = tiny_case
    plain = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"], flags=fg.FLAG_NO_MORTON)
    hip, _ = regs
    tn = _tnodes(np.random.default_rng(4), 16, 0.25)
    rn = fg.RotNode(-0.3, 0.1, 0.2, 0.125)
    a = plain.compute_sse_error(rn, tn, False)
    b = hip.compute_sse_error(rn, tn, False)
    assert rel(a[1], b[1]) <= REL
    plain.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_exact_sse_parity(regs, fg, seed):
    hip, orc = regs
    rng = np.random.default_rng(seed)
    R = fg.synth.random_rotation(rng, 60.0).astype(np.float32)
    t = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    a, b = hip.compute_sse_error(R, t), orc.compute_sse_error(R, t)
    assert abs(float(a) - float(b)) <= REL * float(b)


def test_procrustes_step_parity(regs, fg, tiny_case):
    hip, orc = regs
    rng = np.random.default_rng(12)
    R0 = fg.synth.random_rotation(rng, 15.0).astype(np.float32)
    w = (tiny_case["pcs"] @ R0.T + rng.uniform(-0.05, 0.05, 3)).astype(np.float32)
    R, t, cen, ABt, idx = hip.procrustes(w)
    Ro, to, ceno, ABto, idxo = orc.procrustes(w)
    assert np.array_equal(idx, idxo)                       # nearest neighbours incl. the sqrt tie rule
    assert np.allclose(cen, ceno, rtol=1e-6, atol=1e-7)
    assert np.allclose(ABt, ABto, rtol=1e-5, atol=1e-5)
    assert np.allclose(R, Ro, atol=2e-6) and np.allclose(t, to, atol=2e-6)


def test_nn_tie_rule_lowest_index(fg, oracle, gpu_required):
    """Duplicate target points and exact distance ties: the first index must win (icp3d.cu:20-25)."""
    rng = np.random.default_rng(1)
    base = rng.uniform(-1, 1, size=(300, 3)).astype(np.float32)
    tgt = np.concatenate([base, base[::-1], base])          # every point three times
    src = (base[:200] + rng.normal(scale=1e-3, size=(200, 3))).astype(np.float32)
    bounds = np.array([[-1, 1]] * 3, np.float32)
    hip = fg.Registration(tgt, src, bounds, 0.1)
    orc = oracle.Registration(tgt, src, bounds, 0.1, build_lut=False)
    *_, idx = hip.procrustes(src)
    *_, idxo = orc.procrustes(src)
    assert np.array_equal(idx, idxo)
    assert np.all(idx < 300)
    hip.close()


@pytest.mark.parametrize("thr,angle", [(0.05, 10.0), (0.005, 25.0), (0.0005, 5.0)])
def test_icp_parity(regs, fg, thr, angle):
    hip, orc = regs
    rng = np.random.default_rng(int(angle))
    R0 = fg.synth.random_rotation(rng, angle).astype(np.float32)
    t0 = rng.uniform(-0.05, 0.05, 3).astype(np.float32)
    icp = fg.IterativeClosestPoint3D(hip, None, None, 100, thr, R0, t0)
    sse, R, t = icp.run()
    sse_o, R_o, t_o, it_o = orc.icp(R0, t0, 100, thr)
    assert icp.iterations == it_o
    assert abs(float(sse) - float(sse_o)) <= 1e-5 * float(sse_o)
    assert np.allclose(R, R_o, atol=1e-5) and np.allclose(t, t_o, atol=1e-5)


def test_icp_zero_iterations(regs, fg):
    hip, _ = regs
    icp = fg.IterativeClosestPoint3D(hip, None, None, 0, 0.05, np.eye(3), np.zeros(3))
    sse, R, t = icp.run()
    # max_iter = 0: loop body never runs, sse(1e10) < last_sse(2e10) → returns (1e10, R0, t0)  (icp3d.cu:94,106)
    assert float(sse) == pytest.approx(1e10) and np.array_equal(R, np.eye(3, dtype=np.float32))


def test_invalid_arguments_fail_loudly(fg, gpu_required):
    with pytest.raises(fg.FgoicpError):
        fg.Registration(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), np.array([[0, 1]] * 3, np.float32), -1.0)


# ---- exact BVH nearest neighbour vs the brute-force kernels (both HIP), at sizes the CPU oracle cannot reach ----
@pytest.mark.parametrize("name,res", [("small", 0.02), ("bunny", 0.02)])
def test_bvh_nn_equals_brute_force_at_scale(fg, gpu_required, name, res):
    tgt, src, _, _ = fg.synth.workload(name, angle_deg=60.0)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    tree = fg.Registration(pct, pcs, bounds, res)
    brute = fg.Registration(pct, pcs, bounds, res, flags=fg.FLAG_BRUTE_FORCE_NN)
    # LUT nodes: bit-exact
    a, b = tree.lut_read(), brute.lut_read()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    rng = np.random.default_rng(0)
    for k, shift in enumerate([0.0, 0.3, 3.0]):  # near, off, and far outside the target's box
        R = fg.synth.random_rotation(rng, 90.0).astype(np.float32)
        t = (rng.uniform(-0.1, 0.1, 3) + shift).astype(np.float32)
        # exact SSE: same per-point minima, same summation order -> identical bits
        assert tree.compute_sse_error(R, t).view(np.uint32) == brute.compute_sse_error(R, t).view(np.uint32)
        w = (pcs @ R.T + t).astype(np.float32)
        Ra, ta, cena, ABta, idxa = tree.procrustes(w)
        Rb, tb, cenb, ABtb, idxb = brute.procrustes(w)
        assert np.array_equal(idxa, idxb)
        assert np.array_equal(Ra, Rb) and np.array_equal(ta, tb)
    tree.close(); brute.close()


def test_brute_force_flag_matches_oracle(fg, oracle, tiny_case, gpu_required):
    c = tiny_case
    hip = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"], flags=fg.FLAG_BRUTE_FORCE_NN)
    orc = oracle.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    assert np.array_equal(hip.lut_read().view(np.uint32), orc.lut_get().view(np.uint32))
    w = (c["pcs"] + np.float32(0.03)).astype(np.float32)
    *_, idx = hip.procrustes(w)
    *_, idxo = orc.procrustes(w)
    assert np.array_equal(idx, idxo)
    hip.close()


def test_sorted_and_plain_bounds_paths_agree(fg, tiny_case, gpu_required, monkeypatch):
    """The locality-sorted whole-tick kernel and the plain per-rotation-node kernel compute the same
    per-point values; only the fp64 chunking differs (256- vs 256*P-point chunks)."""
    c = tiny_case
    rng = np.random.default_rng(8)
    nodes = [fg.RotNode(0.5, 0.5, -0.5, 0.5), fg.RotNode(0.0, 0.125, 0.0, 0.25), fg.RotNode(-0.25, 0.25, 0.25, 0.0625)]
    groups = [_tnodes(rng, 32, 0.5), _tnodes(rng, 5, 0.25), _tnodes(rng, 70, 0.125)]
    fixes = [True, False, False]
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FGOICP_BOUNDS_SORTED", mode)
        reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
        out[mode] = reg.compute_bounds_multi([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
        again = reg.compute_bounds_multi([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
        for (a, b), (a2, b2) in zip(out[mode], again):
            assert np.array_equal(a, a2) and np.array_equal(b, b2)  # run-to-run bit-reproducible in either mode
        reg.close()
    for (lb1, ub1), (lb0, ub0) in zip(out["1"], out["0"]):
        assert rel(ub1, ub0) <= REL
        assert np.max(np.abs(lb1.astype(np.float64) - lb0)) <= REL * max(float(np.max(ub0)), 1e-30)


@pytest.mark.parametrize("window", [300, 4096, 0])
def test_bounds_multi_many_groups_and_windows(fg, tiny_case, gpu_required, monkeypatch, window):
    """More subcubes than one window and many rotation nodes: windows split mid-group (window 0 = the default size)."""
    c = tiny_case
    if window:
        monkeypatch.setenv("FGOICP_MAX_SUBCUBES", str(window))
    reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    rng = np.random.default_rng(10)
    nodes = [fg.RotNode(*rng.uniform(-0.4, 0.4, 3), 0.125) for _ in range(40)]
    groups = [_tnodes(rng, int(rng.integers(1, 260)), 0.25) for _ in nodes]
    assert sum(len(g) for g in groups) > 4096
    fixes = [bool(i % 2) for i in range(len(nodes))]
    multi = reg.compute_bounds_multi([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
    for k in (0, 7, 23, 39):
        lb1, ub1 = reg.compute_sse_error(nodes[k], groups[k], fixes[k])
        assert np.array_equal(multi[k][0], lb1) and np.array_equal(multi[k][1], ub1)
    reg.close()


@pytest.mark.parametrize("chunk", [256, 512, 1024, 2048])
def test_work_item_sizes_agree_with_the_oracle(fg, oracle, tiny_case, gpu_required, monkeypatch, chunk):
    """Dense clouds use bigger work items in the sorted bounds kernel (512..2048 points per item): same per-point values,
    only the fp64 partial sums are cut differently."""
    c = tiny_case
    monkeypatch.setenv("FGOICP_CHUNK_PTS", str(chunk))
    reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    orc = oracle.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    rng = np.random.default_rng(21)
    rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)
    tn = _tnodes(rng, 50, 0.25)
    for fix in (True, False):
        lb, ub = reg.compute_sse_error(rn, tn, fix)
        lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
        assert rel(ub, ubo) <= REL
        assert np.max(np.abs(lb.astype(np.float64) - lbo)) <= REL * max(float(np.max(ubo)), 1e-30)
    reg.close()


@pytest.mark.parametrize("size", ["tiny", "bunny"])
def test_small_tick_path_equals_sorted_path(fg, oracle, tiny_case, gpu_required, monkeypatch, size):
    """Small ticks skip the descriptor copies and the locality sort (descriptors read from pinned host memory, items in
    submission order); big ticks are sorted.  Same partial sums either way: bit-identical bounds, and both match the oracle.
    At bunny size the sorted tick has 20k items on all eight XCDs: a wrong permutation out of the device sort (it counts with
    XCD-private L2 atomics) would leave partial sums unwritten and show here."""
    if size == "tiny":
        c = tiny_case
    else:
        tgt, src, *_ = fg.synth.workload("bunny", angle_deg=30.0)
        pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
        c = dict(pct=pct, pcs=pcs, bounds=bounds, res=0.02)
    rng = np.random.default_rng(33)
    rn = fg.RotNode(-0.125, 0.25, 0.125, 0.25)
    tn = _tnodes(rng, 128 if size == "bunny" else 64, 0.125)
    out = {}
    for items in ("0", "100000000"):  # never small / always small
        monkeypatch.setenv("FGOICP_SMALL_TICK", items)
        reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
        out[items] = [reg.compute_sse_error(rn, tn, fix) for fix in (True, False)]
        again = [reg.compute_sse_error(rn, tn, fix) for fix in (True, False)]  # the sort's scratch is back in its initial state
        for (a, b), (a2, b2) in zip(out[items], again):
            assert np.array_equal(a, a2) and np.array_equal(b, b2)
        reg.close()
    for k, fix in enumerate((True, False)):
        (lb0, ub0), (lb1, ub1) = out["0"][k], out["100000000"][k]
        assert np.array_equal(lb0, lb1) and np.array_equal(ub0, ub1)
    if size == "tiny":
        orc = oracle.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
        for k, fix in enumerate((True, False)):
            lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
            assert rel(out["0"][k][1], ubo) <= REL


@pytest.mark.parametrize("workload,res", [("tiny", 0.05), ("bunny", 0.02)])
def test_icp_loop_variants_are_bit_identical(fg, gpu_required, monkeypatch, workload, res):
    """Ways to run IterativeClosestPoint3D::run (icp3d.cu:88-107), all on the same kernels' arithmetic:
      "1"        ONE walk per iteration serves the exact SSE of iteration k and the correspondence pass of iteration k+1 (nn_scan_dual_kernel:
                 the default beyond 262 144 points and in trimmed runs), the reductions that follow started in its epilogue and folded on the
                 host in the device's order;
      dual_unfused  the same with separate reduction kernels;
      two_scans  two scans on two streams (FGOICP_ICP_DUAL=0), fused reductions, every iteration enqueued after its SVD — the default up to
                 262 144 points;
      gated      the same kernels enqueued one iteration AHEAD behind stream gates (hipStreamWaitValue64) the host opens once it has written the
                 motion into pinned memory (FGOICP_ICP_GATED=1; measured slower, a knob);
      unfused    two scans, separate reduction kernels (FGOICP_ICP_FUSE=0);
      sequential one stream, separate kernels (FGOICP_ICP_OVERLAP=0);
      device     the loop advanced ON THE DEVICE (FGOICP_ICP_DEVICE=1: SVD, compose and loop test in a one-thread kernel — the host's
                 SVD source compiled for the device —, passes enqueued ahead; measured slower, kept as a knob).
    Every output bit equal, including the iteration count."""
    tgt, src, R_gt, t_gt = fg.synth.workload(workload, angle_deg=30.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    rng = np.random.default_rng(3)
    out = {}
    for mode in ("1", "dual_unfused", "two_scans", "gated", "unfused", "device", "0"):
        monkeypatch.setenv("FGOICP_ICP_GATED", "1" if mode == "gated" else "0")
        monkeypatch.setenv("FGOICP_ICP_OVERLAP", "0" if mode == "0" else "1")
        monkeypatch.setenv("FGOICP_ICP_DEVICE", "1" if mode == "device" else "0")
        monkeypatch.setenv("FGOICP_ICP_DUAL", "1" if mode in ("1", "dual_unfused") else "0")  # "gated": two scans, fused, pre-enqueued behind stream gates
        monkeypatch.setenv("FGOICP_ICP_FUSE", "0" if mode in ("unfused", "dual_unfused", "0") else "1")
        reg = fg.Registration(pct, pcs, bounds, res)
        runs = []
        for thr, ang, max_iter in ((0.05, 40.0, 100), (0.005, 15.0, 100), (0.0005, 3.0, 100), (0.0, 25.0, 3), (0.005, 20.0, 1), (1e-7, 2.0, 40)):  # 4th, 5th, 6th end on max_iter
            R0 = fg.synth.random_rotation(np.random.default_rng(int(ang)), ang).astype(np.float32)
            t0 = np.array([0.01, -0.02, 0.005], np.float32)
            icp = fg.IterativeClosestPoint3D(reg, None, None, max_iter, thr, R0, t0)
            sse, R, t = icp.run()
            runs.append((np.float32(sse).view(np.uint32), R.copy(), t.copy(), icp.iterations))
        # a Procrustes step and an SSE after ICP runs: the scratch buffers are back in a consistent state
        w = (pcs @ R.T + t).astype(np.float32)
        runs.append(reg.procrustes(w)[4].copy())
        runs.append(np.float32(reg.compute_sse_error(R, t)).view(np.uint32))
        out[mode] = runs
        reg.close()
    for other in ("dual_unfused", "two_scans", "gated", "unfused", "device", "0"):
        for a, b in zip(out["1"][:6], out[other][:6]):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3], (other, a, b)
        assert np.array_equal(out["1"][6], out[other][6]) and out["1"][7] == out[other][7]
    assert out["1"][3][3] == 3 and out["1"][4][3] == 1  # ran into max_iter: the loop's last move of the working cloud has no pass to ride on


@pytest.mark.parametrize("window", [0, 32])
def test_twin_subcubes_are_evaluated_once_with_identical_sums(fg, tiny_case, gpu_required, monkeypatch, window):
    """fgoicp_bounds_submit_twins: a translation node held by the fix_rot group AND the non-fix_rot group of one rotation
    is evaluated with one lookup per point and both variants of the formulae — bit-identical to two evaluations; a wrong
    hint (different node, same fix_rot, other rotation) is ignored."""
    import ctypes as C
    c = tiny_case
    lib = fg._lib.load()
    if window:  # windows of 32 subcubes: the pairs (5..16, 23..34) straddle window borders and fall back to two evaluations
        monkeypatch.setenv("FGOICP_MAX_SUBCUBES", str(window))
    reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    rng = np.random.default_rng(12)
    rn = fg.RotNode(0.125, -0.25, 0.375, 0.25)
    other = fg.RotNode(-0.375, 0.125, 0.25, 0.25)
    ta, tb, tc = _tnodes(rng, 20, 0.25), _tnodes(rng, 24, 0.25), _tnodes(rng, 6, 0.5)
    tb[3:15] = ta[5:17]  # 12 common nodes between group 0 (fix_rot) and group 1 (not)
    R9 = np.concatenate([fg.nodes.to_glm(n.q.R) for n in (rn, rn, other)]).astype(np.float32)
    spans = np.array([rn.span, rn.span, other.span], np.float32)
    fix = np.array([1, 0, 0], np.int32)
    offs = np.array([0, 20, 44, 50], np.int32)
    tn4 = np.ascontiguousarray(np.concatenate([ta, tb, tc]), np.float32)
    twin = np.full(50, -1, np.int32)
    for k in range(12):
        twin[5 + k] = 20 + 3 + k
        twin[20 + 3 + k] = 5 + k
    twin[0], twin[45] = 45, 0      # wrong: other rotation
    twin[1], twin[2] = 2, 1        # wrong: same group / same fix_rot, different node
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    def run(tw):
        lb, ub = np.zeros(50, np.float32), np.zeros(50, np.float32)
        fn = lib.fgoicp_bounds_submit_twins
        rc = fn(reg._h, 0, 3, R9.ctypes.data_as(fp), spans.ctypes.data_as(fp), fix.ctypes.data_as(ip), offs.ctypes.data_as(ip), tn4.ctypes.data_as(fp),
                tw.ctypes.data_as(ip) if tw is not None else None)
        assert rc == 0
        assert lib.fgoicp_bounds_collect(reg._h, 0, lb.ctypes.data_as(fp), ub.ctypes.data_as(fp)) == 0
        return lb, ub
    reg.set_profile(True)
    reg.profile(reset=True)
    lb0, ub0 = run(None)
    p0 = reg.profile(reset=True)
    lb1, ub1 = run(twin)
    p1 = reg.profile(reset=True)
    assert np.array_equal(lb0, lb1) and np.array_equal(ub0, ub1)
    assert p0["subcubes"] == p1["subcubes"] == 50 and p0["evaluations"] == 50
    # 12 pairs evaluated once; with 32-subcube windows only the 9 pairs with both members in the first window (rows 23..31)
    assert p1["evaluations"] == (38 if not window else 41)
    # and they are the bounds of the synchronous per-node operator
    l, u = reg.compute_sse_error(rn, tb, False)
    assert np.array_equal(l, lb1[20:44]) and np.array_equal(u, ub1[20:44])
    reg.close()


# ---- size-independent properties at the benchmark's full size (the CPU oracle cannot reach it) ----
@pytest.fixture(scope="module")
def bunny_full(fg, gpu_required):
    tgt, src, R_gt, t_gt = fg.synth.workload("bunny", angle_deg=150.0, min_angle_deg=110.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    reg = fg.Registration(pct, pcs, bounds, 0.005)
    yield dict(reg=reg, pct=pct, pcs=pcs, bounds=bounds, R_gt=R_gt, t_gt=t_gt, off_t=off_t, off_s=off_s, scale=scale)
    reg.close()


def test_full_size_bounds_are_additive_over_the_source(fg, bunny_full):
    """Bounds are sums over source points: bounds(A u B) = bounds(A) + bounds(B) for a split of the cloud."""
    c = bunny_full
    rng = np.random.default_rng(0)
    mask = rng.random(len(c["pcs"])) < 0.37
    ra = fg.Registration(c["pct"], c["pcs"][mask], c["bounds"], 0.005)
    rb = fg.Registration(c["pct"], c["pcs"][~mask], c["bounds"], 0.005)
    rn = fg.RotNode(-0.375, 0.125, 0.25, 0.125)
    tn = _tnodes(rng, 64, 0.125)
    for fix in (True, False):
        lb, ub = c["reg"].compute_sse_error(rn, tn, fix)
        la, ua = ra.compute_sse_error(rn, tn, fix)
        lbb, ubb = rb.compute_sse_error(rn, tn, fix)
        assert np.allclose(ub, ua.astype(np.float64) + ubb, rtol=2e-6)
        assert np.allclose(lb, la.astype(np.float64) + lbb, rtol=2e-6, atol=2e-6 * float(ub.max()))
    ra.close(); rb.close()


def test_full_size_bound_orderings(fg, bunny_full):
    """lb <= ub; widening the translation cube can only lower lb; the rotation slack can only lower both."""
    reg = bunny_full["reg"]
    rng = np.random.default_rng(1)
    rn = fg.RotNode(0.125, 0.375, -0.125, 0.25)
    t = rng.uniform(-0.5, 0.5, (32, 3)).astype(np.float32)
    prev_lb = None
    for span in (0.0625, 0.125, 0.25, 0.5):
        tn = np.concatenate([t, np.full((32, 1), span, np.float32)], axis=1)
        lb0, ub0 = reg.compute_sse_error(rn, tn, True)
        lb1, ub1 = reg.compute_sse_error(rn, tn, False)
        assert np.all(lb0 <= ub0) and np.all(lb1 <= ub1)
        assert np.all(ub1 <= ub0 * (1 + 1e-6)) and np.all(lb1 <= lb0 * (1 + 1e-6) + 1e-6)
        if prev_lb is not None:
            assert np.all(lb0 <= prev_lb * (1 + 1e-6) + 1e-6)
        prev_lb = lb0


def test_full_size_sse_and_icp_properties(fg, bunny_full):
    c = bunny_full
    reg = c["reg"]
    # ground truth in the scaled frame: R_gt, t' = (t_gt + R_gt c_s - c_t) * scale  (inverse of restore_translation)
    R = c["R_gt"].astype(np.float32)
    t = ((c["t_gt"] + c["R_gt"] @ (-c["off_s"].astype(np.float64)) + c["off_t"].astype(np.float64)) * float(c["scale"])).astype(np.float32)
    sse_gt = float(reg.compute_sse_error(R, t))
    assert sse_gt / len(c["pcs"]) < 2e-4                     # aligned: only sampling + noise residual
    assert float(reg.compute_sse_error(np.eye(3), np.zeros(3))) > 50 * sse_gt
    # the exact-NN scan equals the brute-force kernels at full size (bits of the fp32 result)
    brute = fg.Registration(c["pct"], c["pcs"], c["bounds"], 0.02, flags=fg.FLAG_BRUTE_FORCE_NN)
    assert reg.compute_sse_error(R, t).view(np.uint32) == brute.compute_sse_error(R, t).view(np.uint32)
    brute.close()
    # ICP from the ground truth converges in a few iterations, does not get worse, and is a fixed point
    icp = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.0005, R, t)
    sse1, R1, t1 = icp.run()
    assert float(sse1) <= sse_gt * (1 + 1e-6) and icp.iterations <= 10
    icp2 = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.0005, R1, t1)
    sse2, R2, t2 = icp2.run()
    assert float(sse2) <= float(sse1) * (1 + 1e-6) and np.allclose(R2, R1, atol=1e-4) and icp2.iterations <= 3


def test_full_size_run_schedules_agree(fg, gpu_required):
    """SERIAL (reference order) and ROUND reach the same optimum on the 40k benchmark pair, default threshold."""
    tgt, src, R_gt, t_gt = fg.synth.workload("bunny", angle_deg=150.0, min_angle_deg=110.0)
    res = {}
    for name, (sched, K) in {"serial": (fg.SCHEDULE_SERIAL, 1), "round": (fg.SCHEDULE_ROUND, 32), "adaptive": (fg.SCHEDULE_ROUND, 0)}.items():
        s = fg.FastGoICP(tgt, src, 0.005, 1e-3, schedule=sched, round_width=K)
        R, t = s.run()
        res[name] = (R, t, float(s.get_best_error()), s.stats())
        s.close()
    (Rs, ts, es, _) = res["serial"]
    for other in ("round", "adaptive"):
        Rr, tr, er, _ = res[other]
        assert er == pytest.approx(es, rel=1e-5) and np.allclose(Rs, Rr, atol=1e-5) and np.allclose(ts, tr, atol=1e-5 * max(1.0, float(np.abs(ts).max())))
    ang = np.degrees(np.arccos(np.clip((np.trace(Rs.astype(np.float64).T @ R_gt) - 1) / 2, -1, 1)))
    assert ang < 0.5 and np.linalg.norm(ts - t_gt) < 1e-3


def test_submit_collect_two_slots_equal_synchronous_call(fg, tiny_case, gpu_required):
    """fgoicp_bounds_submit / _collect on both slots at once give the bounds of the synchronous call."""
    import ctypes as C
    from fgoicp_amd import _lib
    from fgoicp_amd.registration import _fp
    c = tiny_case
    reg = fg.Registration(c["pct"], c["pcs"], c["bounds"], c["res"])
    rng = np.random.default_rng(11)
    lib = reg._lib
    subs = []
    for slot in (0, 1):
        nodes = [fg.RotNode(*rng.uniform(-0.4, 0.4, 3), 0.125) for _ in range(6 + slot)]
        groups = [_tnodes(rng, int(rng.integers(1, 900)), 0.25) for _ in nodes]  # slot 1 exceeds one window (4096)
        fixes = [bool(i % 2) for i in range(len(nodes))]
        Rg = np.concatenate([fg.to_glm(n.q.R) for n in nodes]).astype(np.float32)
        spans = np.array([n.span for n in nodes], np.float32)
        fr = np.array([int(f) for f in fixes], np.int32)
        offs = np.zeros(len(nodes) + 1, np.int32); offs[1:] = np.cumsum([len(g) for g in groups])
        tn = np.ascontiguousarray(np.concatenate(groups))
        subs.append((nodes, groups, fixes, Rg, spans, fr, offs, tn))
        _lib.check(lib.fgoicp_bounds_submit(reg._h, slot, len(nodes), _fp(Rg), _fp(spans), fr.ctypes.data_as(_lib.c_int_p),
                                            offs.ctypes.data_as(_lib.c_int_p), _fp(tn)), "fgoicp_bounds_submit")
    # a busy slot refuses a second submission
    assert lib.fgoicp_bounds_submit(reg._h, 0, 0, None, None, None, None, None) != 0
    outs = []
    for slot in (1, 0):
        n = int(subs[slot][6][-1])
        lb = np.empty(n, np.float32); ub = np.empty(n, np.float32)
        _lib.check(lib.fgoicp_bounds_collect(reg._h, slot, _fp(lb), _fp(ub)), "fgoicp_bounds_collect")
        outs.append((slot, lb, ub))
    assert lib.fgoicp_bounds_collect(reg._h, 0, _fp(lb), _fp(ub)) != 0  # nothing in flight any more
    for slot, lb, ub in outs:
        nodes, groups, fixes, *_ , offs, tn = subs[slot]
        ref = reg.compute_bounds_multi([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
        for g, (lbr, ubr) in enumerate(ref):
            assert np.array_equal(lb[offs[g]:offs[g + 1]], lbr) and np.array_equal(ub[offs[g]:offs[g + 1]], ubr)
    reg.close()


@pytest.mark.parametrize("trim", [0.0, 0.25])
def test_icp_batch_equals_single_runs(fg, gpu_required, trim):
    """fgoicp_icp_batch: several ICP runs share the device on their own lanes (scratch, streams, host thread); every run is bit
    for bit the run fgoicp_icp does alone — more runs than lanes, trimmed and untrimmed."""
    tgt, src, R_gt, t_gt = fg.synth.workload("small", angle_deg=40.0)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    reg = fg.Registration(pct, pcs, bounds, 0.02)
    if trim:
        reg.set_inliers(int(len(pcs) * (1 - trim)))
    rng = np.random.default_rng(4)
    Rs = [fg.synth.random_rotation(rng, 50.0).astype(np.float32) for _ in range(7)]
    ts = rng.uniform(-0.2, 0.2, (7, 3)).astype(np.float32)
    sse, Ro, to, it = fg.icp_batch(reg, Rs, ts, 100, 0.005)
    for i in range(7):
        icp = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.005, Rs[i], ts[i])
        e, R, t = icp.run()
        assert e.view(np.uint32) == sse[i].view(np.uint32) and np.array_equal(R, Ro[i]) and np.array_equal(t, to[i]) and icp.iterations == it[i]
    assert len(set(it.tolist())) > 1  # runs of different length were in flight together
    e0, *_ = fg.icp_batch(reg, [], np.zeros((0, 3), np.float32))
    assert e0.size == 0
    reg.close()


def test_context_reports_what_it_derived_from_the_cloud_statistics(fg, gpu_required):
    """fgoicp_ctx_get_info: a sparse cloud (few source points per voxel of the LUT's faces) gets the (apron-bricked) yz-quad LUT copy and 256-point
    items, a dense one the z-pair copy and bigger items (measured crossovers)."""
    rng = np.random.default_rng(2)
    bounds = np.array([[-1, 1]] * 3, np.float32)
    tgt = rng.uniform(-0.9, 0.9, (4000, 3)).astype(np.float32)
    sparse = fg.Registration(tgt, rng.uniform(-0.9, 0.9, (3000, 3)).astype(np.float32), bounds, 0.02)   # 100^3 nodes, 3000 points
    dense = fg.Registration(tgt, rng.uniform(-0.9, 0.9, (60000, 3)).astype(np.float32), bounds, 0.02)   # 2 points per face voxel
    a, b = sparse.info(), dense.info()
    for i, reg in ((a, sparse), (b, dense)):
        assert i["lut_dims"] == reg.lut_dims() and i["lut_nodes"] == int(np.prod(i["lut_dims"]))
        assert i["items_per_evaluation"] == -(-reg.ns // i["points_per_item"]) and i["max_subcubes_per_window"] >= 32
        assert i["source_order"] == 2 and i["tree_order"] == 1  # k-d cells (round 3 defaults; FGOICP_FLAG_CURVE_ORDER / FGOICP_POINT_CURVE / FGOICP_BVH_ORDER change them)
        assert i["lut_bytes"] >= i["lut_nodes"] * 4 * (1 + (4 if i["lut_layout"] in (2, 4) else 2))
    assert a["source_points_per_face_voxel"] == pytest.approx(3000 / 3e4, rel=0.05) and a["lut_layout"] == 4 and a["points_per_item"] == 256
    assert b["source_points_per_face_voxel"] == pytest.approx(2.0, rel=0.05) and b["lut_layout"] == 1 and b["points_per_item"] == 2048
    sparse.close(); dense.close()



def _lattice_tick(fg, rng, n_groups):
    """Groups as the inner BnB submits them (fgoicp.cpp:157-168): whole sibling octets of lattice nodes, mixed with stray nodes, some
    octets incomplete (7 of 8) and one with a wrong span — only genuine octets may be grouped, every row must keep its bits."""
    Rs, spans, fixes, groups = [], [], [], []
    for gi in range(n_groups):
        v = rng.uniform(-0.5, 0.5, 3)
        node = fg.RotNode(*v, float(rng.choice([0.25, 0.125, 0.0625])))
        rows = []
        for _ in range(int(rng.integers(1, 5))):
            span = float(rng.choice([0.25, 0.125, 0.0625]))
            k = rng.integers(-3, 4, 3) * 2 + 1          # parent centre: odd multiples of the parent span 2 * span
            parent = k * (2 * span) * 0.5
            octet = [[parent[0] - span + (j & 1) * 2 * span, parent[1] - span + ((j >> 1) & 1) * 2 * span, parent[2] - span + ((j >> 2) & 1) * 2 * span, span] for j in range(8)]
            octet = [octet[j] for j in rng.permutation(8)]  # a heap pops equal keys in any order
            kind = rng.integers(0, 4)
            if kind == 1:
                octet = octet[:7]                       # incomplete
            if kind == 2:
                octet[3][3] = span * 2                  # one sibling with another span
            rows += octet
            if rng.random() < 0.5:
                rows.append([*rng.uniform(-0.4, 0.4, 3), span])  # a stray node between octets
        Rs.append(node.q.R); spans.append(node.span); fixes.append(bool(gi % 2)); groups.append(np.asarray(rows, np.float32) * np.float32(0.35))
    return Rs, spans, fixes, groups


@pytest.mark.parametrize("trim", [False, True])
@pytest.mark.parametrize("workload,res,chunk", [("tiny", 0.05, None), ("small", 0.02, "1024")])
def test_sibling_units_keep_every_bit(fg, gpu_required, monkeypatch, workload, res, chunk, trim):
    """FGOICP_UNITS = 4 / 8: the children of one translation node share the point loads and the rotation (bounds_units_kernel);
    FGOICP_LDS_TILES = 128 / 192: the LUT brick under a pass of 256 points is staged in LDS and the footprints are read from there
    (bounds_lds_kernel).  Lookups, per-point expressions and the order of every sum are those of the one-evaluation kernel — all bounds
    bit-identical, trimmed or not."""
    tgt, src, R_gt, t_gt = fg.synth.workload(workload, angle_deg=30.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    if chunk:
        monkeypatch.setenv("FGOICP_CHUNK_PTS", chunk)
    monkeypatch.setenv("FGOICP_SMALL_TICK", "0")  # through the sort even for this small tick
    out = {}
    for units in ("0", "4", "8", "lds128", "lds192"):
        monkeypatch.setenv("FGOICP_UNITS", units if units.isdigit() else "0")
        monkeypatch.setenv("FGOICP_LDS_TILES", units[3:] if units.startswith("lds") else "0")  # LUT tiles staged in LDS (bounds_lds_kernel)
        reg = fg.Registration(pct, pcs, bounds, res)
        if trim:
            reg.set_inliers(int(0.8 * len(pcs)))
        args = _lattice_tick(fg, np.random.default_rng(5), 24)
        out[units] = reg.compute_bounds_multi(*args)
        reg.close()
    for units in ("4", "8", "lds128", "lds192"):
        for (lb0, ub0), (lb1, ub1) in zip(out["0"], out[units]):
            assert np.array_equal(lb0.view(np.uint32), lb1.view(np.uint32)) and np.array_equal(ub0.view(np.uint32), ub1.view(np.uint32)), units


@pytest.mark.parametrize("trim", [False, True])
@pytest.mark.parametrize("workload,res", [("tiny", 0.05), ("small", 0.02), ("small", 0.013)])
def test_packed_lut_layouts_keep_every_bit(fg, gpu_required, monkeypatch, workload, res, trim):
    """FGOICP_LUT_ZPAIR = 0 plain / 1 z-pair / 2 yz-quad runs / 3 2x2x2 quad bricks / 4 apron-bricked quads (4 x 2 quads per line,
    lines overlapping by one x): the same texels in the same blend order under every layout — all bounds bit-identical, trimmed or
    not (the bricked layouts have no trimmed kernel and fall back to the runs there; the apron layout has one).  LUT dims that are
    and are not multiples of the brick sizes."""
    tgt, src, R_gt, t_gt = fg.synth.workload(workload, angle_deg=30.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    monkeypatch.setenv("FGOICP_SMALL_TICK", "0")  # through the sort even for this small tick
    out = {}
    for layout in ("2", "0", "1", "3", "4"):
        monkeypatch.setenv("FGOICP_LUT_ZPAIR", layout)
        reg = fg.Registration(pct, pcs, bounds, res)
        if layout == "4":
            assert reg.info()["lut_layout"] == 4
        if trim:
            if layout == "3":
                reg.close()
                continue
            reg.set_inliers(int(0.8 * len(pcs)))
        args = _lattice_tick(fg, np.random.default_rng(5), 24)
        out[layout] = reg.compute_bounds_multi(*args)
        reg.close()
    for layout in out:
        for (lb0, ub0), (lb1, ub1) in zip(out["2"], out[layout]):
            assert np.array_equal(lb0.view(np.uint32), lb1.view(np.uint32)) and np.array_equal(ub0.view(np.uint32), ub1.view(np.uint32)), layout


@pytest.mark.parametrize("flags", [0, "noquant"])
@pytest.mark.parametrize("workload,res,chunk", [("tiny", 0.05, "256"), ("small", 0.02, "256"), ("small", 0.013, "1024")])
def test_item_kernel_keeps_every_bit(fg, gpu_required, monkeypatch, workload, res, chunk, flags):
    """bounds_item_kernel (round 4: the shipped bounds kernel — packed fp32, 32-bit texel addressing, per-pass item kinds; csrc/device/bounds_item.hpp)
    against round 3's bounds_sorted_kernel family (development build, FGOICP_BOUNDS_ITEM=0) on the same submission: every bound bit for bit —
    under the z-pair, yz-quad and apron layouts, trimmed or not, twins (dual items) included, multi-pass items, a cloud whose size is not a multiple
    of a pass (the tail pass with missing points), CUDA's 1.8 fixed-point weights on and off (FGOICP_FLAG_NO_WEIGHT_QUANT)."""
    import ctypes as C
    tgt, src, R_gt, t_gt = fg.synth.workload(workload, angle_deg=30.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    pcs = pcs[:len(pcs) - 3]  # never a whole number of passes
    monkeypatch.setenv("FGOICP_SMALL_TICK", "0")
    monkeypatch.setenv("FGOICP_CHUNK_PTS", chunk)
    lib = fg._lib.load()
    rng = np.random.default_rng(21)
    rn = [fg.RotNode(0.125, -0.25, 0.375, 0.25), fg.RotNode(-0.375, 0.125, 0.25, 0.125)]
    ta, tb, tc, td = _tnodes(rng, 20, 0.25), _tnodes(rng, 24, 0.25), _tnodes(rng, 9, 0.125), _tnodes(rng, 7, 0.125)
    tb[3:15] = ta[5:17]  # twelve twins between group 0 (fix_rot) and group 1
    R9 = np.concatenate([fg.nodes.to_glm(n.q.R) for n in (rn[0], rn[0], rn[1], rn[1])]).astype(np.float32)
    spans = np.array([rn[0].span, rn[0].span, rn[1].span, rn[1].span], np.float32)
    fix = np.array([1, 0, 1, 0], np.int32)
    offs = np.array([0, 20, 44, 53, 60], np.int32)
    tn4 = np.ascontiguousarray(np.concatenate([ta, tb, tc, td]), np.float32)
    twin = np.full(60, -1, np.int32)
    for k in range(12):
        twin[5 + k], twin[20 + 3 + k] = 20 + 3 + k, 5 + k
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    out = {}
    for layout in ("1", "2", "4"):
        monkeypatch.setenv("FGOICP_LUT_ZPAIR", layout)
        for trim in (False, True):
            for item in ("1", "0"):
                monkeypatch.setenv("FGOICP_BOUNDS_ITEM", item)
                reg = fg.Registration(pct, pcs, bounds, res, flags=fg.FLAG_NO_WEIGHT_QUANT if flags else 0)
                if trim:
                    reg.set_inliers(int(0.8 * len(pcs)))
                lb, ub = np.zeros(60, np.float32), np.zeros(60, np.float32)
                assert lib.fgoicp_bounds_submit_twins(reg._h, 0, 4, R9.ctypes.data_as(fp), spans.ctypes.data_as(fp), fix.ctypes.data_as(ip), offs.ctypes.data_as(ip),
                                                      tn4.ctypes.data_as(fp), twin.ctypes.data_as(ip)) == 0
                assert lib.fgoicp_bounds_collect(reg._h, 0, lb.ctypes.data_as(fp), ub.ctypes.data_as(fp)) == 0
                out[(layout, trim, item)] = (lb, ub)
                reg.close()
            a, b = out[(layout, trim, "1")], out[(layout, trim, "0")]
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (layout, trim)
            assert float(a[1].max()) > 0
    for trim in (False, True):  # and the layouts among each other
        for layout in ("2", "4"):
            assert np.array_equal(out[("1", trim, "1")][1].view(np.uint32), out[(layout, trim, "1")][1].view(np.uint32))


def test_sibling_units_whole_run(fg, gpu_required, monkeypatch):
    """A whole FastGoICP::run() with and without sibling units: same counters, same result bits (twins and the memo included)."""
    tgt, src, R_gt, t_gt = fg.synth.workload("small", angle_deg=150.0, min_angle_deg=110.0)
    monkeypatch.setenv("FGOICP_CHUNK_PTS", "1024")
    res = {}
    for units in ("0", "4", "8"):
        monkeypatch.setenv("FGOICP_UNITS", units)
        for trim in (0.0, 0.2):
            s = fg.FastGoICP(tgt, src, 0.02, 1e-4, schedule=fg.SCHEDULE_ROUND, round_width=0, trim_fraction=trim)
            R, t = s.run()
            st = s.stats()
            res[(units, trim)] = (np.float32(s.get_best_error()).view(np.uint32), R.copy(), t.copy(), int(st["trans_cubes"]), int(st["rot_cubes"]))
            s.close()
    for units in ("4", "8"):
        for trim in (0.0, 0.2):
            a, b = res[("0", trim)], res[(units, trim)]
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3:] == b[3:], (units, trim, a, b)


@pytest.mark.gpu
def test_serial_look_ahead_changes_no_counter_on_the_device(fg, gpu_required, monkeypatch):
    """SERIAL with and without its look-ahead (driver.hpp: a task's next nodes ride along as phantom rows into its memo) on a cloud big
    enough for the default to switch it on: same pops, pushes, subcubes, operator calls, ICP runs and result bits — and fewer ticks."""
    tgt, src, _, _ = fg.synth.make_pair(20000, 18000, (0.156, 0.152, 0.118), seed=31, angle_deg=150.0, min_angle_deg=110.0)
    out = {}
    for ahead in ("0", "480"):
        monkeypatch.setenv("FGOICP_SERIAL_AHEAD", ahead)
        s = fg.FastGoICP(tgt, src, 0.01, 1e-4, schedule=fg.SCHEDULE_SERIAL)
        R, t = s.run()
        st = s.stats()
        out[ahead] = ([st[k] for k in ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb", "rounds")], R.copy(), t.copy(), np.float32(s.get_best_error()))
        s.close()
    a, b = out["0"], out["480"]
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3].view(np.uint32) == b[3].view(np.uint32)


def test_early_exit_reports_exact_rows_below_the_threshold_and_the_threshold_above(fg, bunny_full):
    """fgoicp_bounds_submit_cut on the benchmark's clouds (157 work items per subcube): a subcube whose lower bound is below its
    group's threshold T comes back bit for bit as without thresholds, every other one as {T, T} — the same answer in every repetition,
    whichever items the kernel happened to skip; +inf and NULL switch the early exit off; twins keep working; the counters say that
    items were skipped at all."""
    reg = bunny_full["reg"]
    rng = np.random.default_rng(77)
    rn = [fg.RotNode(0.125, -0.25, 0.375, 0.25), fg.RotNode(-0.375, 0.125, 0.25, 0.125), fg.RotNode(0.25, 0.25, -0.125, 0.0625)]
    groups = [_tnodes(rng, 260, 0.25), _tnodes(rng, 300, 0.25), _tnodes(rng, 200, 0.125), _tnodes(rng, 240, 0.5)]
    groups[1][40:200] = groups[0][10:170]          # 160 nodes held by the fix_rot group AND the other group of the same rotation
    Rs = [rn[0].q.R, rn[0].q.R, rn[1].q.R, rn[2].q.R]
    spans = [rn[0].span, rn[0].span, rn[1].span, rn[2].span]
    fixes = [True, False, False, True]
    offs = np.concatenate([[0], np.cumsum([len(g) for g in groups])])
    twin = np.full(offs[-1], -1, np.int32)
    for k in range(160):
        twin[10 + k] = offs[1] + 40 + k
        twin[offs[1] + 40 + k] = 10 + k
    exact = reg.compute_bounds_multi(Rs, spans, fixes, groups)
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(exact, reg.compute_bounds_cut(Rs, spans, fixes, groups, None, twin=twin)))
    cut = np.array([np.median(exact[0][0]), np.quantile(exact[1][0], 0.25), np.inf, np.quantile(exact[3][0], 0.1)], np.float32)
    reg.cut_stats(reset=True)
    first = None
    for rep in range(3):
        got = reg.compute_bounds_cut(Rs, spans, fixes, groups, cut, twin=twin if rep != 1 else None, slot=rep & 1)
        for g, ((lb, ub), (lbx, ubx)) in enumerate(zip(got, exact)):
            below = lbx < cut[g]
            assert np.array_equal(lb[below], lbx[below]) and np.array_equal(ub[below], ubx[below]), g
            assert np.all(lb[~below] == cut[g]) and np.all(ub[~below] == cut[g]), g
        assert not np.isinf(cut[2]) or (np.array_equal(got[2][0], exact[2][0]) and np.array_equal(got[2][1], exact[2][1]))
        if first is None:
            first = got
        assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(first, got))
    offered, skipped = reg.cut_stats(reset=True)
    evaluations = 2 * (1000 - 160) + 1000      # two submissions with 160 twin pairs evaluated once, one without the hint
    assert offered > 0 and offered % evaluations == 0 and offered // evaluations > 16   # (work items per evaluation: chunks of the source cloud)
    assert 0 < skipped < offered
    assert reg.cut_stats() == (0, 0)
