#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace (no PMC) of a chain bench.  Usage: bash scripts/trace_chain.sh <tag> <script>
set -u
TAG=$1; SCRIPT=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/$SCRIPT" > "$OUT/trace.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
st = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(st)) if "ds::" in r["Name"]]
with open(out + "/stats.csv", "w") as f:
    f.write("kernel,calls,avg_us,min_us,max_us,pct\n")
    for r in rows:
        f.write('"%s",%s,%.1f,%.1f,%.1f,%s\n' % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
print(open(out + "/stats.csv").read())
PY
