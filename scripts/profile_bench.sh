#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + the HBM PMC passes (FETCH_SIZE, WRITE_SIZE, each in its own run, no other
# trace domain next to --pmc) of ONE bench.py workload on ONE GPU.
# Usage: bash scripts/profile_bench.sh <tag> [bench.py args, e.g. --config cfg5]   -> gpurun_out/prof_<tag>/{kernel_stats.csv,traffic.json,summary.txt}
# Extra SQ passes: PROFILE_SQ=1 (issue / wait / LDS counters: the compute roofline of the chunked regime); PROFILE_HBM=0 skips the two HBM passes.
# bench.py never starts another program from a profiled process: under rocprofv3 it refuses --gpus > 1 and skips the CPU-baseline workers
# (the profiler's preloaded tool has already initialised the GPU in that process); this script refuses --gpus up front as well.
set -u
TAG=${1:-run}; shift || true
prev=""
for a in "$@"; do
  case "$a" in --gpus=*) g=${a#--gpus=};; *) g="";; esac
  [ "$prev" = "--gpus" ] && g=$a
  if [ -n "$g" ] && [ "$g" != "1" ]; then echo "profile_bench.sh: --gpus $g refused: a profiled bench.py must not start child ranks (profile one GPU)" >&2; exit 2; fi
  prev=$a
done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8      # what bench.py asks the runtime for when it is the application: the traced chain then pipelines as the measured one does
export DS_BENCH_SYNTH=white      # white-noise input: same traffic, no torch FFT kernels in the trace
cd /tmp
ARGS="--steps 50 --warmup 5 --min-region-ms 20 --no-cpu-baseline --no-extras $*"
echo "bench.py $ARGS" > "$OUT/command.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
if [ "${PROFILE_HBM:-1}" = "1" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
fi
if [ "${PROFILE_SQ:-0}" = "1" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_lds.log" 2>&1
  rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_CYCLES --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
fi
cd "$ROOT"
python3 scripts/summarize_profile.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
