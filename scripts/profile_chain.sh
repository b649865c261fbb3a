#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace + HBM PMC passes of a chain bench (scripts/bench_cfg4.py or bench_cfg5.py).
# Usage: bash scripts/profile_chain.sh <tag> <script>     -> gpurun_out/prof_<tag>/{stats.csv, traffic.txt}
set -u
TAG=$1; SCRIPT=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/$SCRIPT" > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/$SCRIPT" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/$SCRIPT" > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(st)) if "ds::" in r["Name"]]
with open(out + "/stats.csv", "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,pct\n")
    for r in rows:
        f.write('"%s",%s,%.1f,%.1f,%.1f,%s\n' % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
def pmc(kind):
    f = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % kind, recursive=True)[0]
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ds::" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return d
fe, wr = pmc("fetch"), pmc("write")
with open(out + "/traffic.txt", "w") as f:
    f.write("# per launch, smallest launch of the run (the one-hop / one-block-per-call regime): HBM bytes = FETCH_SIZE*1024*2 (gfx950 16 B/lane correction) + WRITE_SIZE*1024\n")
    for k in sorted(fe, key=lambda k: -min(fe[k])):
        a, b = min(fe[k]) * 1024 * 2, min(wr.get(k, [0])) * 1024
        f.write("%-62s fetch %9.2f MB  write %9.2f MB  total %9.2f MB\n" % (k, a / 1e6, b / 1e6, (a + b) / 1e6))
print(open(out + "/stats.csv").read()); print(open(out + "/traffic.txt").read())
PY
