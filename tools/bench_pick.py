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
    for path, r in walk(d):
        parent = d
        for k in [p for p in path.split("/") if p][:-1]:
            parent = parent[k]
        wall = parent.get("wall_clock_to_optimum_s") if isinstance(parent, dict) else None
        print(json.dumps({"where": path, "wall_s": wall if wall is not None else d.get("wall_clock_to_optimum_s"), "avg_launch_us": round(r["avg_launch_us"], 1), "launches": r["launches"],
                          "evals_per_launch": round(r["evaluations_per_launch"], 1), "bound": r.get("bound"), "frac": r.get("frac")}))
