#!/usr/bin/env python3
"""Condense the rocprofv3 CSV output of scripts/profile_bench.sh into small files (copied into profiles/ by hand):
kernel_stats.csv (per ds:: kernel: calls, avg / median / min / max duration), traffic.json (HBM bytes per launch of every kernel and
per bench step) and the bench line of the traced run.  Usage: summarize_profile.py <prof_dir> <tag>

HBM bytes follow MI355X_MICROARCH.md's HBM section: FETCH_SIZE and WRITE_SIZE count units of 1024 B; on gfx950 FETCH_SIZE reports half
the bytes of a wide (16 B per lane) coalesced read stream, so fetch = FETCH_SIZE * 1024 * 2, write = WRITE_SIZE * 1024."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, tag = sys.argv[1], sys.argv[2]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(d, sub, "**", pat), recursive=True))


def ours(name):
    return "ds::" in name


rows = defaultdict(list)
for f in find("trace", "*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if ours(r["Kernel_Name"]):
            rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(d, "kernel_stats.csv"), "w") as fh:
    fh.write("kernel,calls,avg_us,median_us,min_us,max_us,total_ms\n")
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        fh.write('"%s",%d,%.2f,%.2f,%.2f,%.2f,%.3f\n' % (k[:100], len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3, sum(v) / 1e6))
print("== kernel trace (%s): %s" % (tag, open(os.path.join(d, "command.txt")).read().strip() if os.path.exists(os.path.join(d, "command.txt")) else ""))
print(open(os.path.join(d, "kernel_stats.csv")).read())
for f in find("trace", "*kernel_stats.csv")[:1]:
    print("-- rocprofv3 --stats:", os.path.relpath(f, d))
    print(open(f).read()[:2500])


def pmc(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if ours(r.get("Kernel_Name", "")):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fe, wr = pmc("pmc_fetch"), pmc("pmc_write")
traffic = {"tag": tag, "note": "HBM bytes = FETCH_SIZE*1024*2 (gfx950 wide-read correction) + WRITE_SIZE*1024; mean over the launches of the run", "kernels": {}}
for k in fe:
    f = fe[k].get("FETCH_SIZE", [])
    w = wr.get(k, {}).get("WRITE_SIZE", [])
    if not f:
        continue
    fb = sum(f) / len(f) * 1024 * 2
    wb = (sum(w) / len(w) * 1024) if w else 0.0
    traffic["kernels"][k[:100]] = {"launches": len(f), "fetch_bytes": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb}
if traffic["kernels"]:
    from collections import Counter
    steps = Counter(v["launches"] for v in traffic["kernels"].values()).most_common(1)[0][0]      # the once-per-step kernels
    traffic["steps"] = steps
    traffic["hbm_bytes_per_step"] = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in traffic["kernels"].values()) / steps
    print("== HBM traffic (PMC), %d steps" % steps)
    for k, v in sorted(traffic["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print("%-70s x%5.1f/step  fetch %9.2f MB  write %9.2f MB  total %9.2f MB" % (k[:70], v["launches"] / steps, v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6, v["hbm_bytes_per_launch"] / 1e6))
    print("per step: %.2f MB" % (traffic["hbm_bytes_per_step"] / 1e6))
json.dump(traffic, open(os.path.join(d, "traffic.json"), "w"), indent=1)
compute = {"tag": tag, "note": "SQ / GRBM counters, mean per launch of each kernel (rocprofv3 --pmc passes of scripts/profile_bench.sh with PROFILE_SQ=1; "
           "kernels run one at a time under counter collection); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, "
           "GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_LDS_IDX_ACTIVE over the 256 CUs", "kernels": {}}
for sub in ("pmc_sq", "pmc_lds", "pmc_sq2"):
    acc = pmc(sub)
    for k, cs in acc.items():
        print("== %s: %s" % (sub, k[:90]))
        for c, v in sorted(cs.items()):
            print("   %-28s mean %16.1f  min %16.1f  max %16.1f  n %d" % (c, sum(v) / len(v), min(v), max(v), len(v)))
            ent = compute["kernels"].setdefault(k[:100], {})
            ent[c] = sum(v) / len(v)
            ent["launches"] = len(v)
if compute["kernels"]:
    json.dump(compute, open(os.path.join(d, "compute.json"), "w"), indent=1)
for name in ("trace.log",):
    p = os.path.join(d, name)
    if os.path.exists(p):
        lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
        if lines:
            open(os.path.join(d, "bench_line.json"), "w").write(lines[-1] + "\n")
            print("== bench line of the traced run\n" + lines[-1])
