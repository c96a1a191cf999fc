"""Randomised differential run of the HIP operators against the CPU oracle (not a test: a campaign to find cases the tests miss;
anything it finds becomes a test).  python tools/fuzz_gpu.py [cases] [seed]

Per case: random cloud sizes (1 .. 6000, ragged), random anisotropic target boxes and LUT resolutions (LUT dims 2 .. ~150 per axis),
uniform or surface-like clouds, optional trimming; compares the LUT (bits), batched bounds through the one-node and the whole-tick
paths (many groups, UB / LB pairs sharing nodes = twins), the exact SSE, one Procrustes step (indices equal) and a short ICP."""
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import fgoicp_amd as fg  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402

f32 = np.float32


def make_cloud(rng, n, box, kind):
    if kind == 0:
        p = rng.uniform(-1, 1, (n, 3))
    elif kind == 1:  # on a bumpy sphere
        v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-12
        p = v * (0.8 + 0.15 * np.sin(5 * v[:, :1]) * np.cos(3 * v[:, 1:2]))
    else:  # clustered, with duplicates
        c = rng.uniform(-0.8, 0.8, (max(1, n // 50), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(scale=0.02, size=(n, 3)) * rng.integers(0, 2, (n, 1))
    return (p * box[None, :]).astype(f32)


SCALE = int(os.environ.get("FUZZ_SCALE", "1"))  # > 1: clouds SCALE times larger (several windows per tick, long rows)


def gen_case(rng, idx):
    """Every random draw of one operator case, in the campaign's order.  Pure numpy (no GPU, no oracle): `case_inputs(seed, idx)`
    replays a campaign up to a case so that anything it found can be rebuilt anywhere — round 2's cases 103 and 505 of seed 1
    (tests/test_gpu_edge_cases.py, tests/test_oracle_kat.py)."""
    nt = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 255, 257, 1000, 2049, 6000])) if rng.random() < 0.5 else int(rng.integers(1, 4000))
    ns = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 1025, 4097])) if rng.random() < 0.5 else int(rng.integers(1, 5000))
    if SCALE > 1:
        nt, ns = nt * SCALE + int(rng.integers(0, SCALE)), ns * SCALE + int(rng.integers(0, SCALE))
    box = rng.uniform(0.15, 0.9, 3)
    tgt = make_cloud(rng, nt, box, int(rng.integers(0, 3)))
    src = (make_cloud(rng, ns, box, int(rng.integers(0, 3))) * f32(rng.uniform(0.5, 1.0))).astype(f32)
    lo = tgt.min(0) - f32(rng.uniform(0, 0.05)); hi = tgt.max(0) + f32(rng.uniform(1e-3, 0.05))
    bounds = np.stack([lo, hi], 1).astype(f32)
    res = float(max((hi - lo).max() / rng.integers(2, 150), 2e-3))
    while float(np.prod(np.ceil((hi - lo) / res))) * nt > 6e8:  # the oracle's LUT build is O(nodes * nt) on the CPU
        res *= 1.3
    trim = rng.random() < 0.3 and ns >= 8
    k = int(rng.integers(max(1, ns // 4), ns)) if trim else 0
    desc = f"case {idx}: nt={nt} ns={ns} res={res:.4g} dims~{np.ceil((hi - lo) / res).astype(int).tolist()} trim_k={k}"
    # whole-tick path: G rotation nodes, UB and LB groups of the same node share translation nodes (twins)
    G = int(rng.integers(1, 40))
    Rs, spans, fixes, groups = [], [], [], []
    while len(Rs) < 2 * G:
        v = rng.uniform(-0.6, 0.6, 3)
        node = fg.RotNode(*v, float(rng.choice([0.5, 0.25, 0.125, 0.0625])))
        nb = int(rng.integers(1, 49))
        tn = np.concatenate([rng.uniform(-0.6, 0.6, (nb, 3)), rng.choice([1.0, 0.5, 0.25, 0.0625], (nb, 1))], 1).astype(f32)
        other = tn[rng.random(nb) < 0.6]
        tn_lb = np.concatenate([other, np.concatenate([rng.uniform(-0.6, 0.6, (3, 3)), np.full((3, 1), 0.125)], 1)]).astype(f32)
        for fix, t in ((True, tn), (False, tn_lb)):
            Rs.append(node.q.R); spans.append(node.span); fixes.append(fix); groups.append(t)
    batch_tn = np.concatenate([rng.uniform(-0.5, 0.5, (9, 3)), rng.choice([1.0, 0.25, 0.0625], (9, 1))], 1).astype(f32)
    R = fg.synth.random_rotation(rng, 40.0).astype(f32)
    t = rng.uniform(-0.2, 0.2, 3).astype(f32)
    return dict(desc=desc, tgt=tgt, src=src, bounds=bounds, res=res, k=k, Rs=Rs, spans=spans, fixes=fixes, groups=groups,
                batch_tn=batch_tn, R=R, t=t)


def case_inputs(seed, idx):
    """The inputs of case `idx` of the operator campaign with seed `seed` (replays the draws of the cases before it)."""
    rng = np.random.default_rng(seed)
    for i in range(idx):
        gen_case(rng, i)
    return gen_case(rng, idx)


ICP_ITERS, ICP_THR = 12, 0.005


def check_case(c):
    """One generated case through the HIP operators and the oracle; None, or a description of the first difference."""
    tgt, src, bounds, res, k, desc = c["tgt"], c["src"], c["bounds"], c["res"], c["k"], c["desc"]
    Rs, spans, fixes, groups = c["Rs"], c["spans"], c["fixes"], c["groups"]
    hip = fg.Registration(tgt, src, bounds, res)
    orc = oracle.Registration(tgt, src, bounds, res)
    try:
        assert np.array_equal(hip.lut_read().view(np.uint32), orc.lut_get().view(np.uint32)), "LUT bits"
        if k:
            hip.set_inliers(k); orc.set_inliers(k)
        out = hip.compute_bounds_multi(Rs, spans, fixes, groups)
        for g in range(len(Rs)):
            lbo, ubo = orc.compute_bounds(Rs[g], spans[g], groups[g], fixes[g])
            lb, ub = out[g]
            scale = max(float(np.abs(ubo).max()), 1e-12)
            assert np.allclose(ub, ubo, rtol=2e-6, atol=1e-6 * scale) and np.allclose(lb, lbo, rtol=2e-6, atol=1e-6 * scale), f"tick bounds, group {g} (fix_rot={fixes[g]})"
        if not k:  # the same tick with thresholds (fgoicp_bounds_submit_cut): rows below theirs keep every bit, the others report the threshold
            q = np.random.default_rng(len(Rs) * 7919 + len(src)).uniform(0.0, 1.0, len(Rs))
            cut = np.array([np.inf if qq > 0.9 else np.quantile(out[g][0], qq) for g, qq in enumerate(q)], f32)
            for rep in range(2):
                got = hip.compute_bounds_cut(Rs, spans, fixes, groups, cut, slot=rep)
                for g in range(len(Rs)):
                    below = out[g][0] < cut[g]
                    assert np.array_equal(got[g][0][below], out[g][0][below]) and np.array_equal(got[g][1][below], out[g][1][below]), f"thresholds, group {g}: a row below its threshold changed"
                    assert np.all(got[g][0][~below] == cut[g]) and np.all(got[g][1][~below] == cut[g]), f"thresholds, group {g}: a row at or above its threshold does not report it"
        rn = fg.RotNode(0.2, -0.1, 0.3, 0.25)
        tn = c["batch_tn"]
        for fix in (True, False):
            lb, ub = hip.compute_sse_error(rn, tn, fix)
            lbo, ubo = orc.compute_bounds(rn.q.R, rn.span, tn, fix)
            scale = max(float(np.abs(ubo).max()), 1e-12)
            assert np.allclose(ub, ubo, rtol=2e-6, atol=1e-6 * scale) and np.allclose(lb, lbo, rtol=2e-6, atol=1e-6 * scale), "batch bounds"
        R, t = c["R"], c["t"]
        a, b = float(hip.compute_sse_error(R, t)), float(orc.compute_sse_error(R, t))
        assert abs(a - b) <= 2e-6 * max(abs(b), 1e-12) + 1e-12, f"sse {a} vs {b}"
        if not k:
            w = (src @ R.T + t).astype(f32)
            _, _, cen, _, ix = hip.procrustes(w)
            _, _, ceno, _, ixo = orc.procrustes(w)
            assert np.array_equal(ix, ixo), "correspondence indices"
            assert np.allclose(cen, ceno, rtol=1e-6, atol=1e-7), "centroids"
        sse, Ri, ti = fg.IterativeClosestPoint3D(hip, None, None, ICP_ITERS, ICP_THR, R, t).run()
        sse_o, Ri_o, ti_o, it_o = orc.icp(R, t, ICP_ITERS, ICP_THR)
        assert abs(float(sse) - float(sse_o)) <= 2e-5 * max(abs(float(sse_o)), 1e-10) + 1e-10, f"icp {sse} vs {sse_o}"
    except AssertionError as e:
        return desc + " -> " + str(e)
    except Exception:
        return desc + " -> EXCEPTION " + traceback.format_exc()[-600:]
    finally:
        hip.close()
    return None


def one_case(rng, idx):
    return check_case(gen_case(rng, idx))


BASINS = []  # whole-run cases in which ROUND and SERIAL ended in different basins (counted in the campaign summary, not failed)


def run_case(rng, idx):
    """A whole FastGoICP::run(): SERIAL must walk the oracle's trajectory (every counter equal), ROUND must end within the band
    of the same optimum when the threshold certifies (below the residual), trimmed or not."""
    nt = int(rng.integers(40, 500)); ns = int(rng.integers(20, 300))
    box = rng.uniform(0.3, 0.9, 3)
    tgt = make_cloud(rng, nt, box, 1)
    Rg = fg.synth.random_rotation(rng, 170.0)
    sel = rng.choice(nt, size=min(ns, nt), replace=False)
    src = ((tgt[sel] - rng.uniform(-0.1, 0.1, 3)) @ Rg + rng.normal(scale=2e-3, size=(len(sel), 3))).astype(f32)
    trim = float(rng.choice([0.0, 0.0, 0.1, 0.3]))
    if trim:
        n_out = max(1, int(0.5 * trim * len(src)))
        src[:n_out] = rng.uniform(-1, 1, (n_out, 3)).astype(f32)
    res = float(rng.choice([0.02, 0.04, 0.08]))
    mse = float(rng.choice([1e-3, 3e-4, 1e-4]))
    desc = f"run {idx}: nt={nt} ns={len(src)} res={res} mse={mse} trim={trim}"
    try:
        o = oracle.FastGoICP(tgt, src, res, mse, trim).run()
        s = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_SERIAL, trim_fraction=trim)
        R, t = s.run()
        st = s.stats()
        got = {k: int(st[k]) for k in o["stats"]}
        e = float(s.get_best_error())
        s.close()
        assert got == o["stats"], f"SERIAL counters {got} vs {o['stats']}"
        assert abs(e - float(o["best_sse"])) <= 1e-5 * max(float(o["best_sse"]), 1e-9), f"SERIAL sse {e} vs {o['best_sse']}"
        assert np.allclose(R, o["R"], atol=1e-5), "SERIAL R"
        # the same trajectory with its evaluations dealt over W ranks (round 3: SERIAL on N ranks, driver.hpp run_task_list_sharded; W ranks on
        # device 0 over the in-process transport): every rank must end with the one-GPU record, counter for counter and bit for bit
        W = int(rng.choice([2, 3, 4]))
        m = fg.MultiGoICP(tgt, src, res, mse, devices=[0] * W, transport=fg.TRANSPORT_IN_PROCESS, schedule=fg.SCHEDULE_SERIAL, trim_fraction=trim)
        Rm, tm = m.run()
        for rk in range(W):
            stm = m.stats(rk)
            gm = {k: int(stm[k]) for k in o["stats"]}
            assert gm == got, f"SERIAL on {W} ranks, rank {rk}: counters {gm} vs {got}"
            assert float(m.get_best_error(rk)) == e, f"SERIAL on {W} ranks, rank {rk}: sse {m.get_best_error(rk)} vs {e}"
        assert np.array_equal(Rm, R) and np.array_equal(tm, t), f"SERIAL on {W} ranks: transform differs from the one-GPU run's"
        m.close()
        K = int(rng.choice([0, 1, 3]))
        r = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=K, trim_fraction=trim)
        Rr, tr = r.run()
        er, str_ = float(r.get_best_error()), r.stats()
        r.close()
        # ... and with every subcube evaluated in full (fgoicp_solver_set_early_exit(0)): the same run, counter for counter and bit for bit
        r = fg.FastGoICP(tgt, src, res, mse, schedule=fg.SCHEDULE_ROUND, round_width=K, trim_fraction=trim)
        r.set_early_exit(False)
        Rf, tf = r.run()
        ef, stf = float(r.get_best_error()), r.stats()
        r.close()
        assert ef == er and np.array_equal(Rf, Rr) and np.array_equal(tf, tr), f"ROUND with / without the early exit: sse {er} vs {ef}"
        assert all(int(str_[k]) == int(stf[k]) for k in o["stats"]), "ROUND with / without the early exit: counters differ"

        band = mse * (len(src) if not trim else int(len(src) * (1 - trim))) + 2e-3 * e  # epsilon-optimal + ICP stop band
        if not (er <= e + band and e <= er + band):
            # Not a defect by itself: the search is truncated at rotation span 0.05 / translation span 0.1 (fgoicp.cpp:56, :155) and
            # below that it relies on the ICP trigger `ub < 1.8 * best` (:74) — which depends on the incumbent at the moment a cube
            # is met, i.e. on the order.  Round 2, seed 23, run 43 (23 source points, LUT at 0.02): SERIAL meets the cube of the
            # optimum with best = 0.367 (0.614 < 0.661: refined, 0.0014), ROUND with best = 0.189 (not refined).  Reported, not failed.
            BASINS.append(desc)
            print(f"NOTE {desc}: ROUND sse {er} vs SERIAL {e} (band {band}): schedules ended in different basins", flush=True)
    except AssertionError as ex:
        return desc + " -> " + str(ex)
    except Exception:
        return desc + " -> EXCEPTION " + traceback.format_exc()[-600:]
    return None


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else "ops"
    rng = np.random.default_rng(seed)
    bad = []
    t0 = time.time()
    for i in range(cases):
        r = (run_case if mode == "run" else one_case)(rng, i)
        if r:
            bad.append(r)
            print("FAIL", r, flush=True)
        if i % 10 == 9:
            print(f"[{i + 1}/{cases}] {time.time() - t0:.0f}s, {len(bad)} failures", flush=True)
    print(f"done: {cases} cases, {len(bad)} failures" + (f", {len(BASINS)} runs in which ROUND ended in another basin than SERIAL (loose threshold: see README)" if mode == "run" else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
