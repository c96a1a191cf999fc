"""Golden fixtures (tests/golden/goicp_golden.npz, made by tests/golden/make_golden.py from the
oracle).  CPU: the oracle still reproduces them bit for bit.  GPU: the HIP path, through the
C ABI, reproduces them (bit-exact for LUT / lookups / indices, 1e-6 relative for fp64-accumulated
sums, 1e-5 for ICP results — the north-star tolerance)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "goicp_golden.npz"))
f32 = np.float32


def bits(a):
    return np.ascontiguousarray(a, f32).view(np.uint32)


@pytest.mark.parametrize("pre", ["syn_", "bun_"])
def test_oracle_reproduces_operator_vectors(oracle, pre):
    reg = oracle.Registration(G[pre + "pct"], G[pre + "pcs"], G[pre + "bounds"], float(G[pre + "res"]))
    assert np.array_equal(bits(reg.lut_get()), bits(G[pre + "lut"]))
    assert np.array_equal(bits(reg.lut_search(G[pre + "q"])), bits(G[pre + "q_val"]))
    x, y, z, span = G[pre + "rot_xyz_span"]
    R, _, _ = oracle.rotation(x, y, z)
    assert np.array_equal(R, G[pre + "rot_R"])
    for fix in (0, 1):
        lb, ub = reg.compute_bounds(R, span, G[pre + "tn"], bool(fix))
        assert np.array_equal(bits(lb), bits(G[pre + f"lb_fix{fix}"])) and np.array_equal(bits(ub), bits(G[pre + f"ub_fix{fix}"]))
    assert bits(reg.compute_sse_error(G[pre + "sse_R"], G[pre + "sse_t"])) == bits(G[pre + "sse"])
    Rp, tp, cen, ABt, idx = reg.procrustes(G[pre + "proc_work"])
    assert np.array_equal(idx, G[pre + "proc_idx"]) and np.array_equal(bits(Rp), bits(G[pre + "proc_R"]))
    sse, Ri, ti, it = reg.icp(G[pre + "sse_R"], G[pre + "sse_t"], 100, 0.005)
    assert it == int(G[pre + "icp_iters"]) and bits(sse) == bits(G[pre + "icp_sse"]) and np.array_equal(bits(Ri), bits(G[pre + "icp_R"]))


def test_oracle_reproduces_full_run(oracle):
    pre = "runsyn_"
    g = oracle.FastGoICP(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]))
    r = g.run()
    assert np.array_equal(bits(r["R"]), bits(G[pre + "R"])) and np.array_equal(bits(r["t"]), bits(G[pre + "t"]))
    assert bits(r["best_sse"]) == bits(G[pre + "sse"])
    st = [r["stats"][k] for k in ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb")]
    assert st == list(G[pre + "stats"])


# ------------------------------------------------------------------------------------------------
# GPU side
# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("pre", ["syn_", "bun_"])
def test_hip_reproduces_operator_vectors(fg, gpu_required, pre):
    reg = fg.Registration(G[pre + "pct"], G[pre + "pcs"], G[pre + "bounds"], float(G[pre + "res"]))
    assert np.array_equal(bits(reg.lut_read()), bits(G[pre + "lut"]))
    assert np.array_equal(bits(reg.lut_search(G[pre + "q"])), bits(G[pre + "q_val"]))
    x, y, z, span = G[pre + "rot_xyz_span"]
    rn = fg.RotNode(x, y, z, span)
    assert np.array_equal(rn.q.R, G[pre + "rot_R"])
    for fix in (0, 1):
        lb, ub = reg.compute_sse_error(rn, G[pre + "tn"], bool(fix))
        assert np.allclose(ub, G[pre + f"ub_fix{fix}"], rtol=1e-6, atol=0)
        assert np.allclose(lb, G[pre + f"lb_fix{fix}"], rtol=1e-6, atol=1e-6 * float(G[pre + f"ub_fix{fix}"].max()))
    assert float(reg.compute_sse_error(G[pre + "sse_R"], G[pre + "sse_t"])) == pytest.approx(float(G[pre + "sse"]), rel=1e-6)
    Rp, tp, cen, ABt, idx = reg.procrustes(G[pre + "proc_work"])
    assert np.array_equal(idx, G[pre + "proc_idx"])
    assert np.allclose(Rp, G[pre + "proc_R"], atol=2e-6) and np.allclose(tp, G[pre + "proc_t"], atol=2e-6)
    icp = fg.IterativeClosestPoint3D(reg, None, None, 100, 0.005, G[pre + "sse_R"], G[pre + "sse_t"])
    sse, Ri, ti = icp.run()
    assert icp.iterations == int(G[pre + "icp_iters"])
    assert float(sse) == pytest.approx(float(G[pre + "icp_sse"]), rel=1e-5)
    assert np.allclose(Ri, G[pre + "icp_R"], atol=1e-5) and np.allclose(ti, G[pre + "icp_t"], atol=1e-5)
    reg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pre,schedule,K", [("runsyn_", 0, 1), ("runsyn_", 1, 1), ("runbun_", 0, 1), ("runbun_", 1, 1), ("runbun_", 1, 4)])
def test_hip_full_run_matches_golden(fg, gpu_required, pre, schedule, K):
    """Final (R, t) and residual within 1e-5 relative of the oracle's (north_star); under the SERIAL
    schedule the whole exploration (every counter) is identical as well."""
    s = fg.FastGoICP(G[pre + "tgt"], G[pre + "src"], float(G[pre + "res"]), float(G[pre + "mse"]), schedule=schedule, round_width=K)
    R, t = s.run()
    assert float(s.get_best_error()) == pytest.approx(float(G[pre + "sse"]), rel=1e-5)
    assert np.allclose(R, G[pre + "R"], atol=1e-5)
    assert np.allclose(t, G[pre + "t"], rtol=1e-5, atol=1e-5 * float(np.abs(G[pre + "t"]).max()))
    st = s.stats()
    if schedule == 0:
        got = [st[k] for k in ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb")]
        assert got == list(G[pre + "stats"])
    s.close()
