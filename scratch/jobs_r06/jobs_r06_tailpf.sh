#!/bin/bash
# cfg5: one build's rate at both call lengths + the chain's parity tests (run once per build for an A/B)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white
run() { c=$1; t=$2; shift 2; python3 $R/bench.py --config $c --hops-per-step $t "$@" --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$TAG %-6s T=%-4s %7.3f M frames/s  %9.4f ms per step' % ('$c', '$t', d['value']/1e6, d['ms_per_step']))" | tee -a $O/tail_prefetch_ab.txt; }
for rep in 1 2 3; do
run cfg5 1
run cfg5 625 --steps 4 --warmup 1
done
cd $R && python3 -m pytest tests -q -m gpu -x -k "subband_gsc or cfg5 or chain or aic" 2>&1 | tail -3
