#!/bin/bash
# cfg4 / wpe_nb: utterance groups (DS_CHAIN_PARTS) by call length — the groups were chosen at one block per call
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white
run() { c=$1; t=$2; shift 2; env "$@" python3 $R/bench.py --config $c --hops-per-step $t --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-6s T=%-4s %-20s %7.3f M frames/s  %9.4f ms per step' % ('$c', '$t', '$*', d['value']/1e6, d['ms_per_step']))" | tee -a $O/chain_parts_ab.txt; }
for rep in 1 2; do
for t in 1 4 312; do for q in 1 2 3 4; do run cfg4 $t DS_CHAIN_PARTS=$q; done; done
done
