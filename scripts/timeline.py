#!/usr/bin/env python3
"""Timeline of one steady-state bench step from a rocprofv3 kernel trace: for every ds:: kernel of the step its start offset, duration
and the idle gap since the latest end seen so far.  Usage: timeline.py <prof_dir> [anchor substring of the step's first kernel]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "dcnotch"
rows = []
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ds::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
if len(starts) < 4:
    sys.exit("anchor kernel '%s' not found often enough" % anchor)
# a step in the middle of the last quarter of the run
i0 = starts[-len(starts) // 8 - 2]
i1 = starts[-len(starts) // 8 - 1]
t0 = rows[i0][0]
latest = t0
busy = 0
print("step of %d launches, %.1f us from first start to next step's first start" % (i1 - i0, (rows[i1][0] - t0) / 1e3))
for s, e, name, q, st in rows[i0:i1]:
    gap = (s - latest) / 1e3
    print("%8.1f us  +%7.1f us  gap %6.1f  q%s s%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, st, name[:70]))
    latest = max(latest, e)
durs = {}
for s, e, name, q, st in rows[starts[len(starts) // 2]:]:
    durs.setdefault(name[:70], []).append((e - s) / 1e3)
print("-- second half of the run: mean duration per kernel")
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    print("%9.2f us x %5d  %s" % (sum(v) / len(v), len(v), k))
