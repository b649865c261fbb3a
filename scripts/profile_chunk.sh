#!/bin/bash
# PMC passes of the chunked regime (T = 125 hops per call): where do the wave cycles go?
set -u
TAG=${1:-chunk}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 5 --warmup 1 --hops-per-step 125 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_lds.log" 2>&1
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
