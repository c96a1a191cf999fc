"""Generates tests/golden/goicp_golden.npz from the CPU oracle (oracle/), in this container.

The reference ships no golden vectors and cannot be built here (CUDA/GLM/Eigen absent), so these
vectors pin the ORACLE's behaviour (regression) and give the GPU tests fixed inputs/outputs; they
are not reference outputs (parity unpinned, see oracle/goicp_oracle.hpp).

Inputs: (a) a seeded synthetic pair; (b) a fixed-seed subsample of the Stanford-bunny demo clouds
the reference's test/bunny.toml registers (data/bunny/model_bunny.txt, data_bunny.txt under
/root/reference — read here only, the subsampled points are stored in the fixture as data).

    python tests/golden/make_golden.py
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import fgoicp_amd as fg  # noqa: E402  (synth + node types only; nothing here touches the GPU)
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "goicp_golden.npz")
REF_DATA = "/root/reference/data/bunny"


def load_txt(path):
    with open(path) as f:
        n = int(f.readline())
        a = np.loadtxt(f, dtype=np.float32)
    assert a.shape == (n, 3)
    return a


def operator_vectors(prefix, pct, pcs, bounds, res, out):
    reg = po.Registration(pct, pcs, bounds, res)
    rng = np.random.default_rng(1234)
    out[prefix + "pct"], out[prefix + "pcs"], out[prefix + "bounds"], out[prefix + "res"] = pct, pcs, bounds, np.float32(res)
    out[prefix + "lut"] = reg.lut_get()
    lo, hi = bounds[:, 0], bounds[:, 1]
    q = np.concatenate([rng.uniform(lo, hi, (3000, 3)), rng.uniform(lo - 1, hi + 1, (3000, 3))]).astype(np.float32)
    out[prefix + "q"], out[prefix + "q_val"] = q, reg.lut_search(q)
    rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)
    tn = np.concatenate([rng.uniform(-0.5, 0.5, (32, 3)), rng.choice([0.5, 0.25, 0.125, 0.0625], (32, 1))], 1).astype(np.float32)
    out[prefix + "rot_xyz_span"] = np.array([0.25, -0.125, 0.375, 0.125], np.float32)
    out[prefix + "rot_R"] = rn.q.R
    out[prefix + "tn"] = tn
    for fix in (0, 1):
        lb, ub = reg.compute_bounds(rn.q.R, rn.span, tn, bool(fix))
        out[prefix + f"lb_fix{fix}"], out[prefix + f"ub_fix{fix}"] = lb, ub
    R = fg.synth.random_rotation(rng, 30.0).astype(np.float32)
    t = rng.uniform(-0.1, 0.1, 3).astype(np.float32)
    out[prefix + "sse_R"], out[prefix + "sse_t"] = R, t
    out[prefix + "sse"] = reg.compute_sse_error(R, t)
    work = (pcs @ R.T + t).astype(np.float32)
    Rp, tp, cen, ABt, idx = reg.procrustes(work)
    out[prefix + "proc_work"] = work
    out[prefix + "proc_R"], out[prefix + "proc_t"], out[prefix + "proc_cen"], out[prefix + "proc_ABt"], out[prefix + "proc_idx"] = Rp, tp, cen, ABt, idx
    sse, Ri, ti, it = reg.icp(R, t, 100, 0.005)
    out[prefix + "icp_sse"], out[prefix + "icp_R"], out[prefix + "icp_t"], out[prefix + "icp_iters"] = sse, Ri, ti, np.int32(it)


def full_run(prefix, tgt, src, res, mse, out):
    t0 = time.time()
    g = po.FastGoICP(tgt, src, res, mse)
    r = g.run()
    print(f"{prefix}: full run {time.time() - t0:.1f}s sse={r['best_sse']} stats={r['stats']}")
    out[prefix + "tgt"], out[prefix + "src"] = tgt, src
    out[prefix + "res"], out[prefix + "mse"] = np.float32(res), np.float32(mse)
    out[prefix + "R"], out[prefix + "t"], out[prefix + "t_scaled"], out[prefix + "sse"] = r["R"], r["t"], r["t_scaled"], r["best_sse"]
    out[prefix + "stats"] = np.array([r["stats"][k] for k in ("trans_cubes", "bounds_calls", "rot_cubes", "icp_runs", "icp_iters", "inner_bnb")], np.int64)


def main():
    out = {}
    # (a) synthetic
    tgt, src, R_gt, t_gt = fg.synth.workload("tiny", angle_deg=25.0)
    pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
    operator_vectors("syn_", pct, pcs, bounds, 0.07, out)  # LUT <= 32^3
    # (b) bunny demo clouds, fixed-seed subsample
    rng = np.random.default_rng(20250302)
    model = load_txt(os.path.join(REF_DATA, "model_bunny.txt"))
    data = load_txt(os.path.join(REF_DATA, "data_bunny.txt"))
    model_s = model[np.sort(rng.choice(len(model), 1500, replace=False))]
    data_s = data[np.sort(rng.choice(len(data), 1000, replace=False))]
    pct, pcs, *_, bounds = fg.synth.preprocess(model_s, data_s)
    operator_vectors("bun_", pct, pcs, bounds, 0.07, out)
    # (c) full runs (small enough for the CPU oracle to repeat in seconds)
    full_run("runbun_", model_s[:700], data_s[:450], 0.05, 1e-3, out)
    tgt, src, R_gt, t_gt = fg.synth.make_pair(700, 500, (0.156, 0.152, 0.118), seed=5, angle_deg=120.0, min_angle_deg=100.0)
    full_run("runsyn_", tgt, src, 0.05, 1e-2, out)
    out["runsyn_R_gt"], out["runsyn_t_gt"] = R_gt, t_gt
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
