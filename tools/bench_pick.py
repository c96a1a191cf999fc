"""Reads bench.py's JSON line(s) on stdin and prints, per object that carries a bounds-kernel profile, the few numbers an A/B needs."""
import json
import sys


def walk(d, path=""):
    if isinstance(d, dict):
        if "avg_launch_us" in d:
            yield path, d
        for k, v in d.items():
            yield from walk(v, f"{path}/{k}")


for line in sys.stdin:
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    if d.get("default_threshold_ms_per_step") is not None:
        il = d.get("icp_latency") or {}
        print(json.dumps({"where": "/default_threshold", "ms_per_step": round(d["default_threshold_ms_per_step"], 3), "icp_us_per_iteration": round(il.get("us_per_iteration", 0.0), 2),
                          "icp_iterations": il.get("iterations"), "seconds_icp": il.get("seconds_icp")}))
    for path, r in walk(d):
        parent = d
        for k in [p for p in path.split("/") if p][:-1]:
            parent = parent[k]
        wall = parent.get("wall_clock_to_optimum_s") if isinstance(parent, dict) else None
        print(json.dumps({"where": path, "wall_s": wall if wall is not None else d.get("wall_clock_to_optimum_s"), "avg_launch_us": round(r["avg_launch_us"], 1), "launches": r["launches"],
                          "evals_per_launch": round(r["evaluations_per_launch"], 1), "bound": r.get("bound"), "frac": r.get("frac"), "work_items_evaluated_frac": r.get("work_items_evaluated_frac")}))
