"""The sharded outer BnB (N > 1) on CPU: world_size 2 over gloo.  Every rank must end with the same
optimum as the single-process run, having exchanged once (all-reduce MIN) + once (all-gather) per
expansion round, and the children must really be split between the ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(REPO, "tests", "golden", "goicp_golden.npz"))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(tmp_path, case, K, world, late="0", coop="0", extra_env=None):
    from tests import host_harness
    host_harness.build()  # once, here: the ranks only load it
    prefix = str(tmp_path / f"out_{case}{K}_{world}_{late}_{coop}")
    env = dict(os.environ, OMP_NUM_THREADS="1" if world > 3 else "2", FGOICP_HOST_THREADS="1" if world > 3 else "4", FGOICP_LATE_ICP=late, FGOICP_TEST_COOP=coop,
               FGOICP_COOP_ICP=coop,  # the flow itself is chosen by cloud size (driver.hpp coop()); these clouds are small: forced here
               **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(REPO, "tests", "dist_worker.py"), prefix, case, str(K)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return [np.load(f"{prefix}.rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("case,K", [("runsyn_", 1), ("runbun_", 2), ("runsyn_", 0)])
def test_world_size_2_round_schedule(tmp_path, case, K):
    ranks = launch(tmp_path, case, K, 2)
    a, b = ranks
    # replicated state stays identical on every rank
    assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"]) and a["sse"] == b["sse"]
    assert a["rounds"] == b["rounds"] and a["exchange_calls"] == b["exchange_calls"] == 2 * a["rounds"]
    # the work is sharded: both ranks evaluated rotation cubes, together as many as one process would
    assert a["rot_cubes"] > 0 and b["rot_cubes"] > 0
    assert abs(int(a["rot_cubes"]) - int(b["rot_cubes"])) <= int(a["rounds"])
    # same global optimum as the serial reference trajectory (north_star: 1e-5 relative)
    assert float(a["sse"]) == pytest.approx(float(G[case + "sse"]), rel=1e-5)
    assert np.allclose(a["R"], G[case + "R"], atol=1e-5)
    assert np.allclose(a["t"], G[case + "t"], atol=1e-5 * max(1.0, float(np.abs(G[case + "t"]).max())))


@pytest.mark.parametrize("case,K,world", [("runsyn_", 1, 3)])
def test_late_joining_refinement_keeps_the_ranks_identical_and_the_optimum(tmp_path, case, K, world):
    """FGOICP_LATE_ICP=1 (a knob; measured slower on the 8-rank replay and off by default): a round's triggered ICP runs overlap the next round's bounds work and enter the
    exchange one round late; one more exchange after the loop collects the last round's.  The replicated state must stay identical on
    every rank, and the result is the same optimum — reached through another sequence of incumbents, so it is compared within the
    refinement ICP's own stop band (an iteration that improves the error by < 0.05 % ends it, fgoicp.cpp:22-23), not bit for bit."""
    ranks = launch(tmp_path, case, K, world, late="1")
    for r in ranks[1:]:
        assert np.array_equal(ranks[0]["R"], r["R"]) and np.array_equal(ranks[0]["t"], r["t"]) and ranks[0]["sse"] == r["sse"]
        assert r["rounds"] == ranks[0]["rounds"]
    a = ranks[0]
    assert a["exchange_calls"] == 2 * a["rounds"] + 2  # + the closing exchange
    # epsilon-optimal (epsilon = ns * mse_threshold), like any other exploration order; on these small clouds the optimum is flat
    # (the bunny subsample: sse within 0.2 % at a rotation 0.9 degrees away), so the transform is compared as an angle
    eps = float(G[case + "mse"]) * len(G[case + "src"])
    assert abs(float(a["sse"]) - float(G[case + "sse"])) <= eps + 2e-3 * float(G[case + "sse"])
    ang = np.degrees(np.arccos(np.clip((np.trace(a["R"].astype(np.float64).T @ G[case + "R"].astype(np.float64)) - 1) / 2, -1, 1)))
    assert ang < 2.0, ang


@pytest.mark.parametrize("case,K,world", [("runsyn_", 0, 2), ("runsyn_", 1, 3), ("runbun_", 2, 3)])
def test_cooperative_rounds_reproduce_the_single_process_run(tmp_path, case, K, world):
    """Cooperative rounds (the flow an exchange with a device all-gather switches on): the child bounds are exchanged
    first, then EVERY rank applies the trigger rule of fgoicp.cpp:74-88 to ALL children in the single-process child order, each
    triggered ICP being one run all ranks execute together.  One exchange per round; the replicated state is identical on every
    rank; and — with the tail-batch rule off, so that a task's batches do not depend on which tasks share its rank — the run IS the
    single-process ROUND run: same sequence of incumbents, same (R, t, sse) bit for bit, same rounds and ICP runs."""
    from tests import host_harness as hh
    env = {"FGOICP_TAIL_BATCH": "0"}
    ranks = launch(tmp_path, case, K, world, coop="1", extra_env=env)
    for r in ranks[1:]:
        assert np.array_equal(ranks[0]["R"], r["R"]) and np.array_equal(ranks[0]["t"], r["t"]) and ranks[0]["sse"] == r["sse"]
        assert r["rounds"] == ranks[0]["rounds"] and r["icp_runs"] == ranks[0]["icp_runs"] and r["icp_iters"] == ranks[0]["icp_iters"]
    a = ranks[0]
    assert 0 < a["exchange_calls"] <= a["rounds"]  # one all-gather per round that evaluated children, nothing else on the CPU backend
    assert sum(int(r["rot_cubes"]) for r in ranks) > int(a["rot_cubes"]) > 0
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        one = hh.HostDriver(G[case + "tgt"], G[case + "src"], float(G[case + "res"]), float(G[case + "mse"]), schedule=1, round_width=K).run()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    if K:  # a fixed round width pops K cubes per round whatever the world size -> the very same rounds (adaptive starts at 32 per rank)
        assert np.array_equal(a["R"], one["R"]) and np.array_equal(a["t"], one["t"]) and np.float32(a["sse"]) == np.float32(one["best_sse"])
        assert int(a["rounds"]) == one["stats"]["rounds"] and int(a["icp_runs"]) == one["stats"]["icp_runs"] and int(a["icp_iters"]) == one["stats"]["icp_iters"]
        assert sum(int(r["rot_cubes"]) for r in ranks) == one["stats"]["rot_cubes"]
    assert float(a["sse"]) == pytest.approx(float(G[case + "sse"]), rel=1e-5) and np.allclose(a["R"], G[case + "R"], atol=1e-5)


@pytest.mark.parametrize("case,world,coop", [("runbun_", 2, "1"), ("runsyn_", 3, "0")])
def test_sharded_serial_schedule_is_the_reference_trajectory(tmp_path, case, world, coop):
    """SERIAL on N ranks (driver.hpp run_task_list_sharded): queue, incumbent, cache and commit order replicated, the inner BnBs of
    every speculative evaluation dealt over the ranks, one all-gather of the task outcomes per evaluation.  EVERY rank must end with
    the reference trajectory's counters — the golden SERIAL record, i.e. the oracle's literal driver: subcubes, operator calls,
    rotation cubes, ICP runs and iterations, inner BnBs — and its result, bit for bit; and the evaluations must really have been
    shared (FGOICP_TIMING-free check: the exchange was used, and a replicated control run needs none)."""
    env = {"FGOICP_TEST_SCHEDULE": "0"}
    ranks = launch(tmp_path, case, 0, world, coop=coop, extra_env=env)
    keys = ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb")
    for r in ranks:
        assert [int(r[k]) for k in keys] == list(G[case + "stats"])
        assert np.array_equal(r["R"], ranks[0]["R"]) and np.array_equal(r["t"], ranks[0]["t"]) and r["sse"] == ranks[0]["sse"]
        assert int(r["exchange_calls"]) == int(ranks[0]["exchange_calls"]) > 0
    a = ranks[0]
    assert np.float32(a["sse"]) == np.float32(G[case + "sse"]) and np.array_equal(a["R"], G[case + "R"]) and np.array_equal(a["t"], G[case + "t"])
    if case != "runsyn_":
        return  # (the control below once, on the small case: the CPU suite's time)
    # FGOICP_SERIAL_SHARD=0: every rank walks the trajectory alone (no exchange at all on the CPU backend) — same record
    ranks = launch(tmp_path, case, 1, world, coop=coop, extra_env=dict(env, FGOICP_SERIAL_SHARD="0"))
    for r in ranks:
        assert [int(r[k]) for k in keys] == list(G[case + "stats"]) and int(r["exchange_calls"]) == 0
        assert np.float32(r["sse"]) == np.float32(G[case + "sse"])


def test_world_size_3_uneven_sharding(tmp_path):
    """8 children over 3 ranks (3 + 3 + 2): padded all-gather slots, identical replicated state everywhere."""
    ranks = launch(tmp_path, "runsyn_", 1, 3)
    for r in ranks[1:]:
        assert np.array_equal(ranks[0]["R"], r["R"]) and np.array_equal(ranks[0]["t"], r["t"]) and ranks[0]["sse"] == r["sse"]
    assert sorted(int(r["rot_cubes"]) for r in ranks) == [2, 3, 3]
    assert float(ranks[0]["sse"]) == pytest.approx(float(G["runsyn_sse"]), rel=1e-5)
    assert np.allclose(ranks[0]["R"], G["runsyn_R"], atol=1e-5)


def test_world_size_8_one_child_per_rank(tmp_path, fg):
    """The 8-GPU shape of the benchmark's first round: 8 children, one per rank; then K = 2 -> 16 children, two per rank.
    Exact-copy clouds under a known motion (one unambiguous optimum: with a loose threshold different exploration
    orders may return different epsilon-optimal solutions — 8 ranks triggering 8 ICPs at once usually find a better one)."""
    rng = np.random.default_rng(21)
    R_gt = fg.synth.random_rotation(rng, 150.0, 140.0)
    t_gt = np.array([0.01, -0.02, 0.015])
    for K in (1, 2):
        late = "1" if K == 2 else "0"
        ranks = launch(tmp_path, "kat_", K, 8, late=late)
        for r in ranks[1:]:
            assert np.array_equal(ranks[0]["R"], r["R"]) and np.array_equal(ranks[0]["t"], r["t"]) and ranks[0]["sse"] == r["sse"]
            assert r["rounds"] == ranks[0]["rounds"] and r["exchange_calls"] == 2 * r["rounds"] + (2 if late == "1" else 0)
        assert all(int(r["rot_cubes"]) >= 1 for r in ranks)
        ang = np.degrees(np.arccos(np.clip((np.trace(ranks[0]["R"].astype(np.float64).T @ R_gt) - 1) / 2, -1, 1)))
        assert ang < 0.05 and np.linalg.norm(ranks[0]["t"] - t_gt) < 1e-4 and float(ranks[0]["sse"]) < 1e-6
