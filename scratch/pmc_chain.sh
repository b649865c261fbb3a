#!/bin/bash
# Run ON THE GPU BOX: SQ instruction counters of the cfg5 chain kernels (one rocprofv3 --pmc pass per counter group).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_chain
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 $ROOT/scripts/bench_cfg5.py > $OUT/g$i.log 2>&1
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ds::" in r["Kernel_Name"]:
            d[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(k)
    for c in sorted(d[k]):
        v = d[k][c]
        print("    %-24s min %14.0f  max %14.0f" % (c, min(v), max(v)))
PY
