#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + HBM PMC passes of the bench command.
# Usage: bash scripts/profile_gpu.sh <tag> [bench args...]     -> gpurun_out/prof_<tag>/ + summaries
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 200 --warmup 20 --no-cpu-baseline $*"
# 1) kernel trace + stats (its own run)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
# 2) PMC passes, each in its own run, no tracing domains combined with --pmc besides kernel dispatch rows
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_lds.log" 2>&1
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
