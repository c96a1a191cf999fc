"""Round trip of a tiny tick (one rotation node, B subcubes) through fgoicp_bounds_batch: python tools/tick_latency.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fgoicp_amd as fg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tgt, src, *_ = fg.synth.workload("bunny", angle_deg=150.0, min_angle_deg=110.0)
pct, pcs, *_, bounds = fg.synth.preprocess(tgt, src)
reg = fg.Registration(pct, pcs, bounds, 0.005)
rn = fg.RotNode(0.25, -0.125, 0.375, 0.125)
rng = np.random.default_rng(0)
tn = np.concatenate([rng.uniform(-0.5, 0.5, (B, 3)), np.full((B, 1), 0.125)], axis=1).astype(np.float32)
for _ in range(200): reg.compute_sse_error(rn, tn, False)
n = 3000
t0 = time.perf_counter()
for _ in range(n): reg.compute_sse_error(rn, tn, False)
dt = (time.perf_counter() - t0) / n
print(f"B={B}: {dt * 1e6:.1f} us per synchronous tick (python call included)")
reg.close()
