#!/bin/bash
# cfg4 at both call lengths and the parity tests of the WPE engines (A/B of a kernel build: run once per build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white
run() { c=$1; t=$2; shift 2; python3 $R/bench.py --config $c --hops-per-step $t "$@" --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$TAG %-6s T=%-4s %7.3f M frames/s  %9.4f ms per step' % ('$c', '$t', d['value']/1e6, d['ms_per_step']))" | tee -a $O/wpe_pad_ab.txt; }
for rep in 1 2; do
run cfg4 1
run cfg4 312 --steps 3 --warmup 1
done
cd $R && python3 -m pytest tests -q -m gpu -x -k "wpe or cfg4 or chain" 2>&1 | tail -3
