"""Condenses rocprofv3 --pmc counter_collection CSVs into profiles/bench_pmc.json (the `traffic` that
bench.py reports) and profiles/rNN_bench_pmc_summary.json.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <round_tag> "<command>"

HBM bytes per launch of the dominant kernel, collected and corrected as MI355X_MICROARCH.md §HBM
prescribes: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), both in KiB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so read bytes = 2 * FETCH_SIZE * 1024."""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = ("bounds_sorted_kernel", "bounds_kernel", "nn_scan_kernel", "lut_build_scan_kernel", "lut_build_scan_coarse_kernel")


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            for name in KERNELS:
                if name + "(" in row["Kernel_Name"] or name + "<" in row["Kernel_Name"]:
                    agg[name][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[(name, row["Counter_Name"])] += 1
                    break
    return agg, cnt


def main():
    fetch_dir, write_dir, tag, command = sys.argv[1:5]
    out = {}
    for d in (fetch_dir, write_dir):
        agg, cnt = load(d)
        for k, v in agg.items():
            o = out.setdefault(k, {})
            for c, val in v.items():
                o[c + "_sum"] = val
                o[c + "_dispatches"] = cnt[(k, c)]
    dom = "bounds_sorted_kernel" if "bounds_sorted_kernel" in out else "bounds_kernel"
    b = out[dom]
    rd = 2 * b["FETCH_SIZE_sum"] * 1024 / b["FETCH_SIZE_dispatches"]
    wr = b["WRITE_SIZE_sum"] * 1024 / b["WRITE_SIZE_dispatches"]
    b["read_bytes_per_launch_corrected"] = rd
    b["write_bytes_per_launch"] = wr
    if "TCC_HIT_sum_sum" in b:
        b["l2_hit_rate"] = b["TCC_HIT_sum_sum"] / (b["TCC_HIT_sum_sum"] + b["TCC_MISS_sum_sum"])
    out["_command"] = command
    out["_note"] = ("FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them, separate --pmc passes; gfx950: FETCH_SIZE counts 128-B "
                    "requests at 64 B -> read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM)")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if len(sys.argv) > 5:  # output directory override (the GPU box writes into gpurun_out/, copied to profiles/ afterwards)
        outdir = sys.argv[5]
        os.makedirs(outdir, exist_ok=True)
        json.dump(out, open(os.path.join(outdir, f"{tag}_bench_pmc_summary.json"), "w"), indent=1, sort_keys=True)
        json.dump({"kernel": dom, "hbm_bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                   "l2_hit_rate": b.get("l2_hit_rate"), "source": f"profiles/{tag}_bench_pmc_summary.json ({command})"},
                  open(os.path.join(outdir, "bench_pmc.json"), "w"), indent=1)
        print(open(os.path.join(outdir, "bench_pmc.json")).read())
        return
    json.dump(out, open(os.path.join(repo, "profiles", f"{tag}_bench_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    json.dump({"kernel": dom, "hbm_bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
               "l2_hit_rate": b.get("l2_hit_rate"), "source": f"profiles/{tag}_bench_pmc_summary.json ({command})"},
              open(os.path.join(repo, "profiles", "bench_pmc.json"), "w"), indent=1)
    print(json.dumps(json.load(open(os.path.join(repo, "profiles", "bench_pmc.json"))), indent=1))


if __name__ == "__main__":
    main()
