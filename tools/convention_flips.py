"""How much of the result depends on conventions the oracle had to CHOOSE (CPU only; test infrastructure).

The reference cannot be built here and ships no outputs, so the oracle's choices where the reference inherits behaviour from nvcc
(-fmad contraction), the CUDA texture unit (1.8 fixed-point weights, blend formula), Thrust (reduction order), libdevice (sinf) and
Eigen (JacobiSVD) are unpinned.  This script is the strongest statement this environment allows: it re-runs whole registrations
under every flip of `goicp_oracle::Conventions` and reports, per flip,
  * the final (R, t, sse) against the default conventions (north_star's bar: 1e-5 relative),
  * whether the SERIAL exploration changed (the six counters of the run: subcubes, operator calls, rotation cubes, ICP runs / iterations,
    inner BnBs) — thresholds such as `lb >= best_error` (fgoicp.cpp:151) or `ub < 1.8 best` (:74) can flip on a last-bit change,
  * and, on the operator vectors of the golden fixture, the largest relative change of a bound / an SSE.
Cases: the two full runs of tests/golden/goicp_golden.npz (bunny demo subsample 700/450 at 0.05, synthetic 700/500), and the REAL
test/bunny.toml clouds as the CLI's loader subsamples them (tests/golden/bunny_toml_clouds.npz: 17 891 / 3 037 points, mse 1e-3) at
LUT resolution 0.02 (0.002 is 6e8 nodes: hours for the oracle's O(nodes x nt) build; the resolution changes the bounds, not which
conventions are exercised).

    python tools/convention_flips.py [--quick] [--out profiles/r03_convention_flips.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import pyoracle as po  # noqa: E402

FLIPS = [
    ("default", {}),
    ("R*p: no fma", dict(fma_matvec=0)),
    ("R*p: fma(a,x, b*y) first", dict(fma_matvec=2)),
    ("dist^2 / |p|^2: no fma", dict(fma_dist=0)),
    ("dist^2 / |p|^2: first product fused", dict(fma_dist=2)),
    ("d - 2|p|^2 sin fused (registration.cu:43,51)", dict(fma_rot_sub=1)),
    ("d - sqrt3*span fused (:33,57)", dict(fma_trans_sub=1)),
    ("every device a*b+c fused the other way", dict(fma_matvec=2, fma_dist=2, fma_rot_sub=1, fma_trans_sub=1)),
    ("no device fma at all", dict(fma_matvec=0, fma_dist=0)),
    ("1.8 weights truncated", dict(tex_weight=1)),
    ("weights not quantised", dict(tex_weight=2)),
    ("8-term weighted blend", dict(tex_blend=1)),
    ("Thrust: fp32 pairwise tree", dict(sum_mode=1)),
    ("Thrust: fp32 serial", dict(sum_mode=2)),
    ("sinf +2 ulp", dict(sin_ulps=2)),
    ("sinf -2 ulp", dict(sin_ulps=-2)),
    ("SVD: round 2's two-sided Jacobi", dict(svd_r2_two_sided=1)),
]


def run_case(tgt, src, res, mse):
    g = po.FastGoICP(tgt, src, res, mse)
    g.use_grid(True)  # exact NN through the grid: bit-identical to the literal loops (tests/test_oracle_kat.py), minutes instead of hours
    r = g.run()
    return dict(R=r["R"].astype(np.float64), t=r["t"].astype(np.float64), sse=float(r["best_sse"]), stats=r["stats"])


def operator_vectors(G, prefix):
    reg = po.Registration(G[prefix + "pct"], G[prefix + "pcs"], G[prefix + "bounds"], float(G[prefix + "res"]))
    out = {}
    for fix in (0, 1):
        lb, ub = reg.compute_bounds(G[prefix + "rot_R"], float(G[prefix + "rot_xyz_span"][3]), G[prefix + "tn"], bool(fix))
        out[f"lb{fix}"], out[f"ub{fix}"] = lb.astype(np.float64), ub.astype(np.float64)
    out["sse"] = np.array([float(reg.compute_sse_error(G[prefix + "sse_R"], G[prefix + "sse_t"]))])
    sse, Ri, ti, it = reg.icp(G[prefix + "sse_R"], G[prefix + "sse_t"], 100, 0.005)
    out["icp_sse"] = np.array([float(sse)])
    out["icp_iters"] = it
    out["lut"] = reg.lut_get().astype(np.float64)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="skip the test/bunny.toml clouds (35 s per flip)")
    ap.add_argument("--out", default=os.path.join(REPO, "profiles", "r03_convention_flips.json"))
    args = ap.parse_args()
    G = np.load(os.path.join(REPO, "tests", "golden", "goicp_golden.npz"))
    cases = [("golden runbun 700/450 @0.05", G["runbun_tgt"], G["runbun_src"], float(G["runbun_res"]), float(G["runbun_mse"])),
             ("golden runsyn 700/500 @0.05", G["runsyn_tgt"], G["runsyn_src"], float(G["runsyn_res"]), float(G["runsyn_mse"]))]
    if not args.quick:
        Z = np.load(os.path.join(REPO, "tests", "golden", "bunny_toml_clouds.npz"))
        cases.append(("test/bunny.toml clouds 17891/3037 @0.02", Z["tgt"], Z["src"], 0.02, float(Z["mse_threshold"])))
    base_runs, base_ops, rows = {}, {}, []
    for name, flip in FLIPS:
        po.reset_conventions()
        po.set_conventions(**flip)
        row = dict(flip=name, conventions=flip, cases={}, operators={})
        t0 = time.time()
        for pre in ("syn_", "bun_"):
            ops = operator_vectors(G, pre)
            if name == "default":
                base_ops[pre] = ops
            b = base_ops[pre]
            rel = lambda a, c: float(np.max(np.abs(a - c) / np.maximum(np.abs(c), 1e-30)))  # noqa: E731
            row["operators"][pre] = dict(bounds_rel=max(rel(ops[k], b[k]) for k in ("lb0", "ub0", "lb1", "ub1")), sse_rel=rel(ops["sse"], b["sse"]),
                                         icp_sse_rel=rel(ops["icp_sse"], b["icp_sse"]), icp_iters=[ops["icp_iters"], b["icp_iters"]],
                                         lut_nodes_changed=int(np.count_nonzero(ops["lut"] != b["lut"])))
        for cname, tgt, src, res, mse in cases:
            r = run_case(tgt, src, res, mse)
            if name == "default":
                base_runs[cname] = r
            b = base_runs[cname]
            row["cases"][cname] = dict(
                sse=r["sse"], sse_rel=abs(r["sse"] - b["sse"]) / b["sse"], R_maxabs=float(np.abs(r["R"] - b["R"]).max()),
                t_rel=float(np.abs(r["t"] - b["t"]).max() / max(np.abs(b["t"]).max(), 1e-30)), counters_equal=r["stats"] == b["stats"],
                counters=r["stats"])
        rows.append(row)
        worst = max(max(c["sse_rel"], c["R_maxabs"], c["t_rel"]) for c in row["cases"].values())
        moved = [c for c, v in row["cases"].items() if not v["counters_equal"]]
        print(f"{name:45s} worst (sse rel, |dR|, t rel) = {worst:.2e}  counters moved in {len(moved)}/{len(cases)}  [{time.time() - t0:.0f}s]", flush=True)
    po.reset_conventions()
    with open(args.out, "w") as f:
        json.dump(dict(cases=[c[0] for c in cases], rows=rows, baseline={k: dict(sse=v["sse"], stats=v["stats"]) for k, v in base_runs.items()}), f, indent=1)
    # markdown table for DESIGN.md
    print("\n| flip | " + " | ".join(c[0] for c in cases) + " | operator vectors: bounds / SSE (max rel) |")
    print("|---|" + "---|" * (len(cases) + 1))
    for row in rows[1:]:
        cells = []
        for c in cases:
            v = row["cases"][c[0]]
            cells.append(f"{max(v['sse_rel'], v['R_maxabs'], v['t_rel']):.1e}" + ("" if v["counters_equal"] else " (counters moved)"))
        ops = row["operators"]
        cells.append(f"{max(o['bounds_rel'] for o in ops.values()):.1e} / {max(o['sse_rel'] for o in ops.values()):.1e}")
        print(f"| {row['flip']} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
