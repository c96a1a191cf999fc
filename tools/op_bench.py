"""Operator-level timing of the bounds kernel on a FIXED tick (results do not feed back into a search, so builds that change the
arithmetic — tools/ablate.sh — stay comparable):  python tools/op_bench.py [workload] [groups] [reps]
G random rotation nodes (span 0.125) x 32 translation nodes (span 0.0625) scattered within +-0.15 of the ground-truth translation in
the scaled frame; fix_rot alternates.  Prints ns per evaluation and the algorithmic GB/s of the kernel.
OP_BENCH_CUT=q: the tick goes through fgoicp_bounds_submit_cut with every group's threshold at the q-quantile of its exact lower bounds
(q = 2: thresholds nothing reaches — the early exit's bookkeeping alone; development build: FGOICP_CUT_PROBE takes it apart)."""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import fgoicp_amd as fg  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "bunny"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
tgt, src, R_gt, t_gt = fg.synth.workload(wl, angle_deg=150.0, min_angle_deg=110.0)
pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
t_scaled = float(scale) * (R_gt @ (-off_s.astype(np.float64)) + t_gt + off_t.astype(np.float64))
reg = fg.Registration(pct, pcs, bounds, 0.005)
rng = np.random.default_rng(0)
nodes, groups, fixes = [], [], []
while len(nodes) < G:
    v = rng.uniform(-1, 1, 3)
    if np.linalg.norm(v) > 0.95:
        continue
    nodes.append(fg.RotNode(*v, 0.125))
    tn = np.concatenate([t_scaled[None, :] + rng.uniform(-0.15, 0.15, (32, 3)), np.full((32, 1), 0.0625)], axis=1).astype(np.float32)
    groups.append(tn)
    fixes.append(bool(len(nodes) % 2))
args = ([n.q.R for n in nodes], [n.span for n in nodes], fixes, groups)
exact = reg.compute_bounds_multi(*args)  # warm-up
q = os.environ.get("OP_BENCH_CUT")
cut = None
if q is not None:
    cut = np.array([1e30 if float(q) > 1 else np.quantile(e[0], float(q)) for e in exact], np.float32)
reg.set_profile(True); reg.profile(reset=True); reg.cut_stats(reset=True)
for _ in range(reps):
    out = reg.compute_bounds_multi(*args) if cut is None else reg.compute_bounds_cut(*args, cut)
p = reg.profile(reset=True)
p["work_items"], p["work_items_not_evaluated"] = reg.cut_stats(reset=True)
ns_eval = p["kernel_ms"] * 1e6 / p["evaluations"]
print(json.dumps({"workload": wl, "groups": G, "cut_quantile": q, "probe": os.environ.get("FGOICP_CUT_PROBE"), "work_items": p["work_items"], "not_evaluated": p["work_items_not_evaluated"],
                  "evaluations": p["evaluations"], "launches": p["launches"], "kernel_us_per_launch": p["kernel_ms"] * 1e3 / p["launches"], "ns_per_evaluation": ns_eval,
                  "algorithmic_GBps": reg.ns * 32.375 / ns_eval, "checksum_ub": float(np.sum([o[1].astype(np.float64).sum() for o in out]))}))
reg.close()
