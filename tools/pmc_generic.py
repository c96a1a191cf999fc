"""Sums every counter of rocprofv3 --pmc counter_collection CSVs per kernel:  python tools/pmc_generic.py <out.json> <dir> [<dir> ...]"""
import collections, csv, glob, json, os, re, sys
out = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = re.split(r"[<(]", name.replace("void ", "").replace("fgoicp::(anonymous namespace)::", ""))[0].strip()
            out[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
res = {k: {c: {"sum": v, "dispatches": cnt[(k, c)], "per_dispatch": v / cnt[(k, c)]} for c, v in cs.items()} for k, cs in out.items()}
json.dump(res, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, b in sorted(res.items()):
    for c, v in sorted(b.items()):
        print(f"{k:32s} {c:24s} sum {v['sum']:.6g}  per dispatch {v['per_dispatch']:.6g}  ({v['dispatches']})")
