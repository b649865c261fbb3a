#!/bin/bash
# cfg4: the two utterance groups half a step apart (DS_CHAIN_STAGGER=1) against in step (default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white
run() { c=$1; t=$2; e=$3; shift 3; env $e python3 $R/bench.py --config $c --hops-per-step $t "$@" --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-6s T=%-4s %-28s %7.3f M frames/s  %9.4f ms per step' % ('$c', '$t', '$e $*', d['value']/1e6, d['ms_per_step']))" | tee -a $O/cfg4_stagger_ab.txt; }
for rep in 1 2; do
run cfg4 312 X=0 --steps 6 --warmup 2
run cfg4 312 DS_CHAIN_STAGGER=1 --steps 6 --warmup 2
run cfg4 1 X=0
run cfg4 1 DS_CHAIN_STAGGER=1 --graph 0
run cfg4 1 X=0 --graph 0
done
