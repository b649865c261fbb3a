#!/usr/bin/env python3
"""The launches of the steady part of a rocprofv3 kernel trace as rows: start, end, duration (us), queue, kernel — to see which
launches wait for which (scripts/concurrency.py gives the totals).  Usage: timeline_rows.py <prof_dir> [n_rows]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ds::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ds::", ""),
                         r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
if not rows:
    sys.exit("no ds:: kernels under %s" % d)
rows.sort()
h = len(rows) // 2
t0 = rows[h][0]
print("%10s %10s %9s  %-6s %-6s %s" % ("start us", "end us", "dur us", "queue", "stream", "kernel"))
for s, e, n, q, st in rows[h:h + n_rows]:
    print("%10.1f %10.1f %9.1f  %-6s %-6s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, n))
