"""Where does the wall time of a step go?  Reads a rocprofv3 --kernel-trace CSV and reports, for the span
between the first and the last bounds kernel: how long the bounds kernel ran, how long ANY kernel ran, and the
idle gaps between consecutive bounds kernels bucketed by length and by what ran inside them.

    python tools/trace_gaps.py <rocprof_out_dir> <out.json> [skip_fraction]

skip_fraction (e.g. 0.5 for a trace of two identical steps): drop that share of the bounds launches from the front, so that the
one-time costs of a first step (code-object loads, first-touch of buffers) stay out of the picture.
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    for k in ("bounds_sorted_kernel", "bounds_kernel", "bounds_finalize", "tick_keys", "tick_scan", "tick_scatter", "nn_scan", "icp_", "sum_", "transform_inplace",
              "trim_select", "copyBuffer", "fillBuffer", "lut_"):
        if k in name:
            return k
    return name[:40]


def main():
    d, outp = sys.argv[1], sys.argv[2]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0),
                         int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
    rows.sort()
    bounds = [r for r in rows if r[2] == "bounds_sorted_kernel"]
    if len(sys.argv) > 3 and bounds:
        bounds = bounds[int(len(bounds) * float(sys.argv[3])):]
        rows = [r for r in rows if r[0] >= bounds[0][0]]
    if not bounds:
        json.dump({"error": "no bounds kernel in trace"}, open(outp, "w"))
        return
    # split into steps: a gap > 20 ms between bounds kernels separates runs (ICP phases are < 20 ms)
    t_first, t_last = bounds[0][0], bounds[-1][1]
    span = t_last - t_first
    busy_bounds = sum(e - s for s, e, *_ in bounds)
    # union of all kernels inside the span
    iv = sorted((max(s, t_first), min(e, t_last)) for s, e, *_ in rows if e > t_first and s < t_last)
    busy_any, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy_any += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy_any += cur_e - cur_s
    # gaps between consecutive bounds kernels
    buckets = collections.OrderedDict((k, [0, 0]) for k in ("<5us", "5-20us", "20-50us", "50-100us", "100-300us", "300us-1ms", "1-20ms", ">20ms"))
    edges = [5e3, 20e3, 50e3, 100e3, 300e3, 1e6, 20e6]
    inside = collections.Counter()
    others = [r for r in rows if r[2] != "bounds_sorted_kernel"]
    oi = 0
    for a, b in zip(bounds, bounds[1:]):
        g = b[0] - a[1]
        if g < 0:
            g = 0
        k = 0
        while k < len(edges) and g >= edges[k]:
            k += 1
        key = list(buckets)[k]
        buckets[key][0] += 1
        buckets[key][1] += g
        while oi < len(others) and others[oi][1] <= a[1]:
            oi += 1
        j = oi
        names = set()
        while j < len(others) and others[j][0] < b[0]:
            names.add(others[j][2])
            j += 1
        tag = "icp" if ("nn_scan" in names or "icp_" in names) else ("sort-only" if names else "nothing")
        inside[tag + " " + key] += g
    # tick sizes: workgroups per bounds launch
    wg = sorted(r[3] // max(r[4], 1) for r in bounds)
    dur_by_size = collections.OrderedDict()
    for lo, hi in ((0, 1000), (1000, 10000), (10000, 50000), (50000, 150000), (150000, 300000), (300000, 10**9)):
        sel = [r for r in bounds if lo <= r[3] // max(r[4], 1) < hi]
        if sel:
            dur_by_size[f"{lo}-{hi} items"] = {"launches": len(sel), "total_ms": sum(e - s for s, e, *_ in sel) / 1e6,
                                               "ns_per_item": sum(e - s for s, e, *_ in sel) / max(1, sum(r[3] // max(r[4], 1) for r in sel))}
    out = {"span_ms": span / 1e6, "bounds_launches": len(bounds), "bounds_busy_ms": busy_bounds / 1e6, "bounds_busy_frac": busy_bounds / span,
           "any_kernel_busy_ms": busy_any / 1e6, "any_kernel_busy_frac": busy_any / span,
           "gaps_between_bounds_kernels": {k: {"count": v[0], "total_ms": v[1] / 1e6} for k, v in buckets.items()},
           "gap_ms_by_content": {k: v / 1e6 for k, v in sorted(inside.items(), key=lambda kv: -kv[1])},
           "items_per_launch_percentiles": {p: wg[min(len(wg) - 1, int(p / 100 * len(wg)))] for p in (5, 25, 50, 75, 95, 100)},
           "bounds_time_by_launch_size": dur_by_size}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
