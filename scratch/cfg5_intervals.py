#!/usr/bin/env python3
"""cfg5 pipeline: distribution of the step length (start-to-start of successive McSpp launches) and mean kernel durations from a rocprofv3 kernel trace.
Usage: cfg5_intervals.py <prof_dir>"""
import csv, glob, os, sys
import statistics as st
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ds::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
mc = [s for s, e, n in rows if "binop_kernel<13" in n]
iv = [(b - a) / 1e3 for a, b in zip(mc[:-1], mc[1:])]
iv = iv[10:]
q = sorted(iv)
print("steps %d: step length us  mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (len(iv), st.mean(iv), q[len(q) // 10], q[len(q) // 2], q[9 * len(q) // 10], q[-1]))
# per 25-step window
print("windows of 25 steps (mean us):", " ".join("%.0f" % st.mean(iv[i:i + 25]) for i in range(0, len(iv) - 24, 25)))
dur = {}
for s, e, n in rows:
    k = n.split("(")[0].replace("void ds::", "")
    dur.setdefault(k, []).append((e - s) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-44s n %4d  mean %7.1f us" % (k[:44], len(v), st.mean(v)))
