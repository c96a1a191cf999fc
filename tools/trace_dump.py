"""Prints the last N rows of a rocprofv3 --kernel-trace CSV as a timeline: start offset (us), duration (us), queue, kernel.
    python tools/trace_dump.py <rocprof_out_dir> [n_rows=120] [name_filter]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
flt = sys.argv[3] if len(sys.argv) > 3 else None
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("fgoicp::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), name[:60]))
rows.sort()
if flt:
    first = next((i for i, r in enumerate(rows) if flt in r[3]), 0)
    rows = rows[first:]
    rows = rows[:n]
else:
    rows = rows[-n:]
t0 = rows[0][0]
prev_end = t0
for s, e, q, name in rows:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} q{q:>3} {name}")
