"""Not a test: microseconds per ICP iteration (IterativeClosestPoint3D::run, icp3d.cu:88-107) for the loop variants.
    python tools/icp_bench.py [workload=bunny] [repeats=5]
Prints JSON lines per variant (csrc/device/ctx.hip): fused reductions (default), separate reduction kernels, the device-resident loop, one stream."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

VARIANTS = {"default": {}, "gated_two_scans_fused": {"FGOICP_ICP_GATED": "1"}, "dual_walk_fused": {"FGOICP_ICP_DUAL": "1"}, "two_scans_unfused": {"FGOICP_ICP_GATED": "0", "FGOICP_ICP_DUAL": "0", "FGOICP_ICP_FUSE": "0"}, "device_loop": {"FGOICP_ICP_DEVICE": "1"},
            "one_stream_unfused": {"FGOICP_ICP_DUAL": "0", "FGOICP_ICP_FUSE": "0", "FGOICP_ICP_OVERLAP": "0"}}


def child(name, repeats):
    import numpy as np
    import fgoicp_amd as fg
    tgt, src, R_gt, t_gt = fg.synth.workload(name, angle_deg=20.0)
    pct, pcs, off_t, off_s, scale, bounds = fg.synth.preprocess(tgt, src)
    reg = fg.Registration(pct, pcs, bounds, 0.005)
    best = None
    for thr in (0.005, 0.0005):
        for _ in range(repeats):
            icp = fg.IterativeClosestPoint3D(reg, None, None, 100, thr, np.eye(3), np.zeros(3))
            t0 = time.perf_counter(); sse, R, t = icp.run(); dt = time.perf_counter() - t0
            r = dict(thr=thr, iters=icp.iterations, ms=dt * 1e3, us_per_iter=dt / max(icp.iterations, 1) * 1e6, sse=float(sse))
            if best is None or (r["thr"] == best["thr"] and r["us_per_iter"] < best["us_per_iter"]) or r["thr"] != best["thr"]:
                if best is not None and r["thr"] != best["thr"]:
                    print(json.dumps(dict(workload=name, ns=len(pcs), variant=os.environ.get("ICP_VARIANT"), **best)), flush=True)
                best = r
    print(json.dumps(dict(workload=name, ns=len(pcs), variant=os.environ.get("ICP_VARIANT"), **best)), flush=True)
    reg.close()


if __name__ == "__main__":
    if os.environ.get("ICP_VARIANT"):
        child(sys.argv[1], int(sys.argv[2]))
    else:
        name = sys.argv[1] if len(sys.argv) > 1 else "bunny"
        rep = sys.argv[2] if len(sys.argv) > 2 else "5"
        for v, env in VARIANTS.items():
            subprocess.run([sys.executable, os.path.abspath(__file__), name, rep], env={**os.environ, **env, "ICP_VARIANT": v}, check=True)
