"""raw window of a rocprofv3 kernel trace: start offset, duration, queue/stream of N consecutive ds:: kernels in the middle of the run,
and mean durations.  usage: trace_window.py <dir> [N]"""
import csv, glob, os, sys
d = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ds::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ds::", "")[:48], r.get("Queue_Id", ""), r.get("Stream_Id", ""), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
i0 = len(rows) * 3 // 4
t0 = rows[i0][0]
for s, e, n, q, st, g in rows[i0:i0 + N]:
    print("%9.1f us +%7.1f us  q%s s%s grid %8d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, g, n))
durs = {}
for s, e, n, q, st, g in rows[len(rows) // 2:]:
    durs.setdefault((n, g), []).append((e - s) / 1e3)
print("-- mean durations, second half")
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    print("%9.2f us x %5d  grid %8d %s" % (sum(v) / len(v), len(v), k[1], k[0]))
