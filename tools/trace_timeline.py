"""Prints a slice of a rocprofv3 --kernel-trace CSV as a timeline (start offset, duration, queue, kernel, workgroups).

    python tools/trace_timeline.py <rocprof_out_dir> <first_bounds_launch> <n_rows>
"""
import csv
import glob
import os
import sys

from trace_gaps import short

d, first, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                     int(r.get("Grid_Size_X", 0) or 0) // max(1, int(r.get("Workgroup_Size_X", 1) or 1))))
rows.sort()
k = 0
start = 0
for i, r in enumerate(rows):
    if r[2] == "bounds_sorted_kernel":
        if k == first:
            start = i
            break
        k += 1
t0 = rows[start][0]
print(f"{'start_us':>10} {'dur_us':>9} {'q':>3} {'s':>3} {'wgs':>8} kernel")
for r in rows[max(0, start - 5):start + n]:
    print(f"{(r[0] - t0) / 1e3:10.1f} {(r[1] - r[0]) / 1e3:9.1f} {r[3]:>3} {r[4]:>3} {r[5]:8d} {r[2]}")
