#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + PMC passes) into small text/JSON summaries
under <dir>/ (copied into profiles/ by hand).  Usage: summarize_prof.py <prof_dir> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]
summary = {"tag": tag}


def find(sub, pat):
    return sorted(glob.glob(os.path.join(d, sub, "**", pat), recursive=True))


# kernel trace: per-kernel count / avg / min / max duration
rows = defaultdict(list)
for f in find("trace", "*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
kstats = {}
for k, v in rows.items():
    v.sort()
    kstats[k] = {"calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": v[0] / 1e3, "max_us": v[-1] / 1e3,
                 "median_us": v[len(v) // 2] / 1e3, "total_ms": sum(v) / 1e6}
summary["kernel_trace"] = kstats
print("== kernel trace (%s)" % tag)
for k, s in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])[:8]:
    print("%-90s calls %5d avg %9.2f us  median %9.2f  min %9.2f  max %9.2f  total %9.3f ms"
          % (k[:90], s["calls"], s["avg_us"], s["median_us"], s["min_us"], s["max_us"], s["total_ms"]))
for f in find("trace", "*kernel_stats.csv"):
    print("-- rocprofv3 --stats file:", os.path.relpath(f, d))
    print(open(f).read()[:1500])

# PMC passes: average counter value per dispatch of our kernel
pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    acc = defaultdict(list)
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ds_frames_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        # skip the warm-up launches' share: plain mean over all launches of the kernel
        pmc[c] = {"mean": sum(v) / len(v), "n": len(v), "min": min(v), "max": max(v)}
summary["pmc"] = pmc
print("== PMC per launch of ds_frames_kernel")
for c, s in sorted(pmc.items()):
    print("%-28s mean %16.1f  min %16.1f  max %16.1f  n %d" % (c, s["mean"], s["min"], s["max"], s["n"]))
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE
    # reports exactly half the bytes of a wide (16 B/lane) coalesced read stream -> double it.
    fetch_raw = pmc["FETCH_SIZE"]["mean"] * 1024.0
    write = pmc["WRITE_SIZE"]["mean"] * 1024.0
    traffic = {"fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected_x2": 2.0 * fetch_raw, "write_bytes": write,
               "hbm_bytes_per_launch": 2.0 * fetch_raw + write, "tag": tag,
               "note": "FETCH_SIZE*1024*2 (gfx950 16B/lane correction) + WRITE_SIZE*1024, mean per launch of ds_frames_kernel"}
    summary["traffic"] = traffic
    json.dump(traffic, open(os.path.join(d, "traffic.json"), "w"), indent=1)
    print("== traffic per launch: fetch(raw) %.2f MB, fetch(x2) %.2f MB, write %.2f MB, total %.2f MB"
          % (fetch_raw / 1e6, 2 * fetch_raw / 1e6, write / 1e6, (2 * fetch_raw + write) / 1e6))
json.dump(summary, open(os.path.join(d, "summary.json"), "w"), indent=1)
