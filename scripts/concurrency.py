#!/usr/bin/env python3
"""How the kernels of a chained workload share the GPU in time, from a rocprofv3 kernel trace (scripts/profile_bench.sh keeps it under
<prof_dir>/trace): over the steady second half of the run —
  * the share of wall time with 0, 1, 2, 3+ ds:: kernels in flight (stream pipelining only pays in the 2+ shares);
  * per kernel: launches, mean duration, and its mean duration split by how many OTHER kernels were in flight for most of it (a kernel that
    takes as long next to another one as the two take one after the other gained nothing from the overlap);
  * the sum of all kernel durations against the wall time (= the average number of kernels in flight).
Usage: concurrency.py <prof_dir>  ->  text on stdout (kept as profiles/<round>/<workload>_concurrency.txt)"""
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ds::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
if not rows:
    sys.exit("no ds:: kernels in %s/trace" % d)
rows.sort()
t_lo = rows[len(rows) // 2][0]
rows = [r for r in rows if r[0] >= t_lo]
t_hi = max(r[1] for r in rows)
wall = t_hi - t_lo
ev = sorted([(s, 1) for s, e, n in rows] + [(e, -1) for s, e, n in rows])
share = {}
depth, last = 0, t_lo
for t, dlt in ev:
    share[min(depth, 3)] = share.get(min(depth, 3), 0) + (t - last)
    depth += dlt
    last = t
print("steady half of the run: %.1f ms of wall time, %d launches" % (wall / 1e6, len(rows)))
print("kernels in flight   share of wall time")
for k in range(4):
    print("  %s            %5.1f %%" % ("3+" if k == 3 else str(k) + " ", 100.0 * share.get(k, 0) / wall))
busy = sum(e - s for s, e, n in rows)
print("sum of kernel durations / wall time = %.2f (average kernels in flight)" % (busy / wall))
# per kernel, by overlap: for each launch the time-weighted mean number of OTHER kernels in flight
import bisect
starts = [r[0] for r in rows]
by = {}
for i, (s, e, n) in enumerate(rows):
    ov = 0
    j = bisect.bisect_left(starts, s - 50_000_000)           # launches that started up to 50 ms earlier can still be running
    for s2, e2, n2 in rows[j:bisect.bisect_right(starts, e)]:
        if s2 == s and e2 == e and n2 == n:
            continue
        ov += max(0, min(e, e2) - max(s, s2))
    others = ov / max(1, e - s)
    by.setdefault(n, []).append(((e - s) / 1e3, others))
print("%-62s %7s %10s %12s %12s %12s" % ("kernel", "calls", "mean us", "alone (<0.25)", "~1 other", ">=1.75 others"))
for n, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    def m(sel):
        xs = [x[0] for x in v if sel(x[1])]
        return "%9.1f(%d)" % (sum(xs) / len(xs), len(xs)) if xs else "        -"
    print("%-62s %7d %10.1f %12s %12s %12s" % (n, len(v), sum(x[0] for x in v) / len(v), m(lambda o: o < 0.25), m(lambda o: 0.25 <= o < 1.75), m(lambda o: o >= 1.75)))
