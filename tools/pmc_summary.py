"""Condenses rocprofv3 --pmc counter_collection CSVs into profiles/bench_pmc.json (the `traffic` bench.py reports per workload)
and profiles/<tag>_bench_pmc_summary.json (every counter of every kernel of interest).

    python tools/pmc_summary.py <tag> <outdir> <read_factor> <leg>=<dir>[,<dir>...] [<leg>=...]

Each <dir> is the -d directory of one `rocprofv3 --pmc ... --kernel-trace --output-format csv -- python3 bench.py --only <leg> ...`
pass (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots).  Units: FETCH_SIZE / WRITE_SIZE are KiB.  Read bytes =
read_factor * FETCH_SIZE * 1024: MI355X_MICROARCH.md §HBM gives 2 for wide coalesced streams on gfx950 (128-byte requests
tallied at 64 B) and asks for a calibration on one's own access pattern otherwise — tools/calib/fetch_calib.hip is that
calibration for divergent 16-byte gathers (result and the factor used: profiles/<tag>_fetch_calibration.txt)."""
import collections
import csv
import glob
import json
import os
import re
import sys

KERNELS = ("bounds_item_kernel", "bounds_sorted_kernel", "bounds_kernel", "trim_rows_sampled_kernel", "trim_rows_kernel", "nn_scan_kernel", "tick_keys_kernel", "tick_scatter_xcd_kernel")


def load(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = re.split(r"[<(]", row["Kernel_Name"].replace("void ", "").replace("fgoicp::(anonymous namespace)::", ""))[0].strip()
                if name in KERNELS:
                    agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[(name, row["Counter_Name"])] += 1
    return agg, cnt


def main():
    tag, outdir, factor = sys.argv[1], sys.argv[2], float(sys.argv[3])
    summary, bench = {}, {}
    for spec in sys.argv[4:]:
        leg, dirs = spec.split("=")
        agg, cnt = load(dirs.split(","))
        ks = {}
        for k, v in agg.items():
            o = ks.setdefault(k, {})
            for c, val in v.items():
                o[c + "_sum"] = val
                o[c + "_dispatches"] = cnt[(k, c)]
            if "FETCH_SIZE_sum" in o:
                o["read_bytes_per_launch"] = factor * o["FETCH_SIZE_sum"] * 1024 / o["FETCH_SIZE_dispatches"]
            if "WRITE_SIZE_sum" in o:
                o["write_bytes_per_launch"] = o["WRITE_SIZE_sum"] * 1024 / o["WRITE_SIZE_dispatches"]
            if "TCC_HIT_sum_sum" in o:
                o["l2_hit_rate"] = o["TCC_HIT_sum_sum"] / (o["TCC_HIT_sum_sum"] + o["TCC_MISS_sum_sum"])
        summary[leg] = ks
        dom = next((k for k in ("bounds_item_kernel", "bounds_sorted_kernel", "bounds_kernel") if k in ks), "bounds_item_kernel")
        if dom in ks and "read_bytes_per_launch" in ks[dom]:
            b = ks[dom]
            e = {"kernel": dom, "hbm_bytes_per_launch": b["read_bytes_per_launch"] + b.get("write_bytes_per_launch", 0.0),
                 "read_bytes_per_launch": b["read_bytes_per_launch"], "write_bytes_per_launch": b.get("write_bytes_per_launch"), "l2_hit_rate": b.get("l2_hit_rate"),
                 "fetch_size_read_factor": factor,
                 "source": f"profiles/{tag}_bench_pmc_summary.json [{leg}]: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 bench.py --only {leg} --steps 1 --warmup 1"}
            sel = next((k for k in ("trim_rows_sampled_kernel", "trim_rows_kernel") if k in ks and "read_bytes_per_launch" in ks[k]), None)
            if sel:
                t = ks[sel]
                e["select_kernel_read_bytes_per_launch"] = t["read_bytes_per_launch"]
                e["select_kernel_write_bytes_per_launch"] = t.get("write_bytes_per_launch")
            bench[leg] = e
    summary["_note"] = ("FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, separate --pmc passes; read bytes = "
                        f"{factor} * FETCH_SIZE * 1024 (see the module docstring of tools/pmc_summary.py)")
    os.makedirs(outdir, exist_ok=True)
    json.dump(summary, open(os.path.join(outdir, f"{tag}_bench_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    json.dump(bench, open(os.path.join(outdir, "bench_pmc.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(bench, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
