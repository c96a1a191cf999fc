"""Host logic of the product (driver template, node types, SVD, pre-processing) without a GPU:
tests/host_harness instantiates fast-go-icp_amd/csrc/host/driver.hpp with the oracle's operators
and the results are compared with the oracle's own literal restatement of fgoicp.cpp."""
import os

import numpy as np
import pytest

from tests import host_harness as hh

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goicp_golden.npz"))
f32 = np.float32
KEYS = ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb")


def test_rotation_and_overlap_match_oracle(oracle):
    rng = np.random.default_rng(0)
    grid = [(-1 + s + i * 2 * s, -1 + s + j * 2 * s, -1 + s + k * 2 * s, s) for s in (0.5, 0.25) for i in range(int(1 / s))
            for j in range(int(1 / s)) for k in range(int(1 / s))]
    pts = grid + [tuple(rng.uniform(-1, 1, 3)) + (float(rng.choice([0.5, 0.25, 0.125, 0.0625])),) for _ in range(200)]
    for x, y, z, s in pts:
        R, r, ok = hh.rotation(x, y, z)
        Ro, ro, oko = oracle.rotation(x, y, z)
        assert ok == oko and np.array_equal(R, Ro) and r == ro
        assert hh.overlaps(x, y, z, s) == oracle.rotnode_overlaps(x, y, z, s)


def test_python_node_types_match_oracle(oracle, fg):
    rng = np.random.default_rng(1)
    for _ in range(200):
        x, y, z = rng.uniform(-1, 1, 3)
        s = float(rng.choice([0.5, 0.25, 0.125, 0.0625]))
        n = fg.RotNode(x, y, z, s)
        Ro, ro, oko = oracle.rotation(x, y, z)
        assert np.array_equal(n.q.R, Ro) and n.q.r == ro and n.q.in_SO3() == oko
        assert n.overlaps_SO3() == oracle.rotnode_overlaps(x, y, z, s)
    a, b = fg.TransNode(0, 0, 0, 0.5, lb=1.0), fg.TransNode(0, 0, 0, 0.25, lb=1.0)
    assert b < a and not (a < b)          # equal lb: the larger span has priority
    assert fg.TransNode(0, 0, 0, 1, lb=2.0) < fg.TransNode(0, 0, 0, 0.1, lb=1.0)  # smaller lb has priority


def test_product_svd_and_procrustes_rotation(oracle):
    """csrc/host/math3.hpp restates Eigen's JacobiSVD algorithm independently of oracle/ (flat code for n = 3 against the
    oracle's rotation objects): U, S, V and the Procrustes rotation must agree BIT FOR BIT on every rank — on rank-deficient H the
    null-space columns are the algorithm's choice, and round 2's Hestenes SVD chose differently (fuzz seed 7, cases 103 / 505)."""
    from tests.test_oracle_kat import _rank_deficient_family
    rng = np.random.default_rng(2)
    for trial in range(800):
        H = _rank_deficient_family(rng, trial)
        U, S, V = hh.svd3(H)
        Uo, So, Vo = oracle.svd3(H)
        assert np.array_equal(U, Uo) and np.array_equal(S, So) and np.array_equal(V, Vo), trial
        scale = max(np.abs(H).max(), 1e-300)
        assert np.allclose(U @ np.diag(S) @ V.T, H, atol=1e-14 * scale) and np.all(np.diff(S) <= 0)
        assert np.allclose(U.T @ U, np.eye(3), atol=1e-13) and np.allclose(V.T @ V, np.eye(3), atol=1e-13)
        ABt = H.T.reshape(9).astype(f32)
        got = hh.closest_orthogonal(ABt)
        assert np.array_equal(got.view(np.uint32), oracle.closest_orthogonal(ABt).view(np.uint32)), trial
        got = got.reshape(3, 3).T
        assert np.allclose(got @ got.T, np.eye(3), atol=1e-5) and np.linalg.det(got.astype(np.float64)) == pytest.approx(1, abs=1e-5)
        if S[2] > 1e-3 * S[0]:  # full rank: unique answer, numpy's LAPACK SVD gives it too
            Un, Sn, Vtn = np.linalg.svd(H)
            want = Vtn.T @ np.diag([1, 1, np.linalg.det(Vtn.T @ Un.T)]) @ Un.T
            assert np.allclose(got, want, atol=1e-5)


def test_preprocessing_matches_oracle(oracle):
    tgt, src = G["runbun_tgt"], G["runbun_src"]
    h = hh.HostDriver(tgt[:200], src[:150], 0.2, 1e-3)
    o = oracle.FastGoICP(tgt[:200], src[:150], 0.2, 1e-3).preproc()
    p = h.preproc()
    for k in ("offset_pcs", "offset_pct", "bounds"):
        assert np.array_equal(p[k], o[k])
    assert p["scale"] == o["scale"]


@pytest.mark.parametrize("pre", ["runsyn_", "runbun_"])
def test_serial_schedule_reproduces_reference_trajectory(pre):
    """Same operators underneath → the product's SERIAL driver must take exactly the oracle
    driver's path: every counter equal, result bit-identical."""
    r = hh.HostDriver(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=0).run()
    assert [r["stats"][k] for k in KEYS] == list(G[pre + "stats"])
    assert np.array_equal(r["R"], G[pre + "R"]) and np.array_equal(r["t"], G[pre + "t"]) and r["best_sse"] == G[pre + "sse"]
    assert np.array_equal(r["t_scaled"], G[pre + "t_scaled"])


@pytest.mark.parametrize("tasks,ahead", [("64", "480"), ("4", "96")])
def test_serial_look_ahead_leaves_the_trajectory_alone(monkeypatch, tasks, ahead):
    """SERIAL look-ahead (driver.hpp prepare_half: the nodes a task pops next ride along as phantom rows into its memo; a batch the
    memo serves entirely is consumed without an operator call).  By default it is off for clouds this small (host-bound runs), so it
    is switched on explicitly: every counter and the result must still be the golden record of the oracle's literal driver."""
    monkeypatch.setenv("FGOICP_SERIAL_AHEAD_TASKS", tasks)
    monkeypatch.setenv("FGOICP_SERIAL_AHEAD", ahead)
    pre = "runsyn_"
    for sched in (0, 5):  # 5 = SERIAL over the two-slot pipelined task loop with the twin-task memo logic on (as on the GPU)
        r = hh.HostDriver(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=sched).run()
        assert [r["stats"][k] for k in KEYS] == list(G[pre + "stats"]), sched
        assert np.array_equal(r["R"], G[pre + "R"]) and np.array_equal(r["t"], G[pre + "t"]) and r["best_sse"] == G[pre + "sse"]


@pytest.mark.parametrize("pre,K,sched", [("runsyn_", 1, 1), ("runsyn_", 3, 1), ("runbun_", 2, 1), ("runsyn_", 2, 2), ("runbun_", 4, 2), ("runbun_", 0, 1), ("runsyn_", 0, 2)])
def test_round_schedule_reaches_the_same_optimum(pre, K, sched):
    """sched 1 = ROUND, synchronous task loop; sched 2 = ROUND with the two-slot pipelined task loop; K = 0: adaptive width."""
    r = hh.HostDriver(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=sched, round_width=K).run()
    assert float(r["best_sse"]) == pytest.approx(float(G[pre + "sse"]), rel=1e-5)
    assert np.allclose(r["R"], G[pre + "R"], atol=1e-5) and np.allclose(r["t"], G[pre + "t"], atol=1e-5 * max(1.0, float(np.abs(G[pre + "t"]).max())))
    if sched == 1:
        assert r["stats"]["bounds_calls"] < G[pre + "stats"][1]  # batches of many tasks share one submission


def test_pipelined_and_synchronous_task_loops_are_equivalent(monkeypatch):
    """A task's own sequence of batches does not depend on how the tasks are grouped into submissions — with the one rule that
    looks at the group switched off (bigger batches for a half with few tasks left, FGOICP_TAIL_BATCH)."""
    monkeypatch.setenv("FGOICP_TAIL_BATCH", "0")
    pre = "runbun_"
    a = hh.HostDriver(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=1, round_width=4).run()
    b = hh.HostDriver(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=2, round_width=4).run()
    assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) and a["best_sse"] == b["best_sse"]
    for k in ("trans_cubes", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb", "rounds"):
        assert a["stats"][k] == b["stats"][k], k
    # two balanced halves: at most two submissions where the synchronous loop needs one
    assert a["stats"]["bounds_calls"] <= b["stats"]["bounds_calls"] <= 2 * a["stats"]["bounds_calls"]
    # With the tail batches ON (ROUND's default) a task's batch size depends on the company it keeps, so the two modes may differ — but
    # only inside the threshold the inner BnB stops on (fgoicp.cpp:120): the same epsilon-optimal answer, not the same bits.
    monkeypatch.delenv("FGOICP_TAIL_BATCH")
    a = hh.HostDriver(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=1, round_width=4).run()
    b = hh.HostDriver(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=2, round_width=4).run()
    eps = float(G[pre + "mse"]) * len(G[pre + "src"])
    assert abs(float(a["best_sse"]) - float(b["best_sse"])) <= eps + 2e-3 * float(a["best_sse"])


def test_twin_task_memo_changes_nothing_a_task_can_see():
    """The memo of the twin task's evaluations (driver.hpp prepare_half: phantom rows for the UB task's nodes, LB batches served
    from the memo, whole batches consumed without a submission): every counter and the result are those of the run without it —
    under expansion rounds (schedule 4 vs 2) and under the reference's order with speculation (5 vs 3, which also equals the
    oracle's literal driver)."""
    pre = "runsyn_"
    args = (G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]))
    for plain, memo, K in ((2, 4, 4), (3, 5, 1)):
        a = hh.HostDriver(*args, schedule=plain, round_width=K).run()
        b = hh.HostDriver(*args, schedule=memo, round_width=K).run()
        assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) and a["best_sse"] == b["best_sse"]
        for k in ("trans_cubes", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb", "rounds"):
            assert a["stats"][k] == b["stats"][k], (plain, k)
    assert [b["stats"][k] for k in KEYS] == list(G[pre + "stats"])  # schedule 5: the oracle's counters


@pytest.mark.parametrize("pre", ["runsyn_", "runbun_"])
def test_early_exit_answers_change_nothing_the_search_can_see(pre):
    """fgoicp_bounds_submit_cut: every inner branch-and-bound hands the operator the value T above which it does not need a subcube's
    exact bounds (driver.hpp InnerTask::cut_above: its running best error; 1.8 x the job's best error for the pass whose result feeds
    the trigger rule), and the device answers {T, T} for such a subcube.  Here the ORACLE operator gives exactly those answers
    (OracleOps::apply_cut) — under every schedule the run must be the run with exact answers: same counters, same bits.  SERIAL is
    moreover the golden record of the oracle's literal restatement of fgoicp.cpp, which knows nothing of thresholds."""
    args = (G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]))
    # (the bunny pair takes 15-30 s per run on the CPU: there only what the drop-in classes run by default — SERIAL over the pipelined task loop with the memo)
    for sched, K in (((0, 1), (3, 1), (5, 1), (1, 3), (2, 4), (4, 0)) if pre == "runsyn_" else ((5, 1),)):
        runs = []
        for passes, applies in ((False, False), (True, True)):
            h = hh.HostDriver(*args, schedule=sched, round_width=K)
            h.set_cut(passes, applies)
            runs.append(h.run())
        a, b = runs
        assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) and a["best_sse"] == b["best_sse"], sched
        assert a["stats"] == b["stats"], sched
        if sched in (0, 3, 5):
            assert [b["stats"][k] for k in KEYS] == list(G[pre + "stats"])
            assert np.array_equal(b["R"], G[pre + "R"]) and np.array_equal(b["t"], G[pre + "t"]) and b["best_sse"] == G[pre + "sse"]


def test_serial_speculation_modes_walk_the_same_trajectory(tmp_path):
    """FGOICP_SERIAL_SPECULATE = 0 (literal, one task at a time), 1 (inside the popped node), 2 (across the tops of the queue,
    default): the same pops, counters and result — speculation only changes how many tasks share an operator submission."""
    import json
    import subprocess
    import sys
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from tests import host_harness as hh; "
            "G = np.load(%r); pre = sys.argv[2]; "
            "r = hh.HostDriver(G[pre + 'tgt'], G[pre + 'src'], float(G[pre + 'res']), float(G[pre + 'mse']), schedule=int(sys.argv[1])).run(); "
            "print(json.dumps({'stats': {k: int(v) for k, v in r['stats'].items()}, 'R': r['R'].tolist(), 't': r['t'].tolist(), 'sse': float(r['best_sse'])}))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goicp_golden.npz")))

    def run(mode, sched, pre):  # sched 3 = SERIAL over the asynchronous (two-slot) operator path of the harness
        env = dict(os.environ, FGOICP_SERIAL_SPECULATE=mode)
        p = subprocess.run([sys.executable, "-c", code, sched, pre], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    for pre, combos in (("runbun_", [("0", "0"), ("2", "0"), ("2", "3")]), ("runsyn_", [("0", "0"), ("1", "0"), ("1", "3"), ("2", "3")])):
        ref = run(*combos[0], pre)
        assert [ref["stats"][k] for k in KEYS] == list(G[pre + "stats"])
        for mode, sched in combos[1:]:
            o = run(mode, sched, pre)
            assert o["stats"] == ref["stats"] and o["R"] == ref["R"] and o["t"] == ref["t"] and o["sse"] == ref["sse"], (pre, mode, sched)


@pytest.mark.parametrize("n,leaf", [(1, 32), (31, 32), (33, 32), (64, 64), (1000, 32), (6000, 64), (40097, 64)])
def test_kd_order_is_a_permutation_whose_runs_are_kd_cells(n, leaf):
    """csrc/device/morton.hpp kd_order: the order the target tree and the source cloud are stored in.  It must be a permutation for
    any size (ragged last leaf, fewer points than a leaf), and every node of the implicit complete binary tree — a run of
    leaf << k points starting at a multiple of its size — must be split by an axis-aligned plane between its two children
    (max of the left <= min of the right along one axis): that is what makes every run's box tight.  Ties, duplicates and
    non-finite coordinates must not break it (the comparator is a total order on the bit patterns)."""
    from tests import host_harness as hh
    rng = np.random.default_rng(n)
    p = rng.normal(size=(n, 3)).astype(np.float32)
    p[rng.integers(0, n, max(1, n // 10))] = p[0]  # duplicates
    perm = hh.point_order(p, leaf, 2, True)
    assert sorted(perm.tolist()) == list(range(n))
    q = p[perm]
    cap = 1
    while cap * leaf < n:
        cap *= 2
    size = cap * leaf
    while size > leaf:
        half = size // 2
        for b in range(0, n, size):
            if b + half >= n:
                continue  # the right child is empty
            left, right = q[b:b + half], q[b + half:min(n, b + size)]
            assert (left.max(0) <= right.min(0)).any(), (size, b)
        size = half
    # the other orders are permutations too, and the density split falls back to the k-d order when nothing is scattered
    for mode in (1, 3):
        assert sorted(hh.point_order(p, leaf, mode, True).tolist()) == list(range(n))
    if n >= 4:
        bad = p.copy()
        bad[1] = np.nan; bad[2] = np.inf; bad[3] = -np.inf
        assert sorted(hh.point_order(bad, leaf, 2, True).tolist()) == list(range(n))


def test_density_split_order_puts_scattered_points_last():
    from tests import host_harness as hh
    rng = np.random.default_rng(5)
    v = rng.normal(size=(20000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    out = rng.uniform(-1.5, 1.5, size=(5000, 3))
    p = np.concatenate([v, out]).astype(np.float32)
    p = p[rng.permutation(len(p))]
    perm = hh.point_order(p, 64, 3, True)
    assert sorted(perm.tolist()) == list(range(len(p)))
    r = np.linalg.norm(p[perm], axis=1)
    on_surface = np.abs(r - 1.0) < 1e-3
    assert on_surface[:19000].mean() > 0.93 and on_surface[-4000:].mean() < 0.05  # an outlier that shares its grid cell counts as dense


def test_leaf_slab_distance_never_exceeds_the_exact_one():
    """csrc/device/slab.hpp: the slab test of the exact scans prunes a leaf when slab_d2(q) exceeds the query's bound, so in
    fp32 it must stay a LOWER bound of the exact squared distance of q from the slab along n — also when the three products of n.q cancel:
    a plane through the origin at 45 degrees, clouds with a large common offset (+100), queries placed right on the pruning boundary.  Exact
    side: numpy longdouble on the same fp32 inputs."""
    from tests.host_harness import slab_d2
    rng = np.random.default_rng(11)
    f32, ld = np.float32, np.longdouble
    worst = 0.0
    cases = []
    for off in (0.0, 1.0, 100.0, 1000.0):
        for _ in range(40):
            n = rng.normal(size=3)
            cases.append((n / np.linalg.norm(n), off))
        cases.append((np.array([1.0, 1.0, 0.0]) / np.sqrt(2.0), off))      # 45 degrees
        cases.append((np.array([1.0, -1.0, 1.0]) / np.sqrt(3.0), off))
    for n, off in cases:
        nf = (n * (1.0 - 1e-6)).astype(f32)                                  # bvh.hip: |n| <= 1 after rounding
        c = rng.uniform(-1, 1, 3) + off                                      # the leaf's centre; off: a cloud far from the origin
        if off == 0.0 and rng.random() < 0.5:
            c = c - n * (n @ c)                                              # a plane through the origin: n.c = 0, the products cancel
        pts = (c + rng.normal(scale=0.01, size=(32, 3)) - n * rng.normal(scale=1e-4, size=(32, 1))).astype(f32)
        pr = (pts.astype(ld) * nf.astype(ld)).sum(axis=1)
        a, b = f32(np.nextafter(f32(pr.min()), f32(-np.inf))), f32(np.nextafter(f32(pr.max()), f32(np.inf)))  # outward, as bvh.hip stores them
        # queries: everywhere, and a batch a hair outside either plane (where an over-estimate flips the decision)
        q_far = (c + rng.normal(scale=2.0, size=(400, 3))).astype(f32)
        t = rng.normal(scale=1.0, size=(400, 3))
        t -= np.outer(t @ n, n)                                              # in-plane offsets
        eps = np.concatenate([rng.uniform(0, 1e-5, 200), rng.uniform(0, 1e-7, 200)]) * max(1.0, off)
        sign = np.where(rng.random(400) < 0.5, 1.0, -1.0)
        q_edge = (c + t + np.outer(sign * (0.02 + eps), n)).astype(f32)
        for qi, q in enumerate((q_far, q_edge)):
            got = slab_d2(nf, a, b, q).astype(ld)
            nq = (q.astype(ld) * nf.astype(ld)).sum(axis=1)
            s = np.maximum(nq - ld(b), ld(a) - nq)
            exact = np.where(s > 0, s * s, ld(0))
            assert np.all(got <= exact), (nf, off, float((got - exact).max()))
            pos = exact > 1e-2  # (far queries only: right at the boundary the allowance rightly leaves nothing)
            if qi == 0 and pos.any():
                worst = max(worst, float(np.max(1 - np.sqrt(got[pos] / exact[pos]))))
    assert worst < 0.05  # the allowance costs the far queries (the ones the slab exists for) less than 5 % of their distance even at offset 1000
    # and it does prune: a query two leaf sizes away along n is still separated
    assert slab_d2(np.array([0, 0, 1], f32) * f32(1 - 1e-6), -0.01, 0.01, np.array([[0.3, 0.2, 0.5]], f32))[0] > 0.2
