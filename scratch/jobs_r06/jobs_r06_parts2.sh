#!/bin/bash
# cfg4 / wpe_nb with 10 s per call: utterance groups (DS_CHAIN_PARTS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white
run() { c=$1; t=$2; shift 2; env "$@" python3 $R/bench.py --config $c --hops-per-step $t --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>$O/err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-6s T=%-4s %-20s %7.3f M frames/s  %9.4f ms per step' % ('$c', '$t', '$*', d['value']/1e6, d['ms_per_step']))" | tee -a $O/chain_parts_ab.txt; }
for rep in 1 2; do
for q in 1 2 3; do run cfg4 312 DS_CHAIN_PARTS=$q; done
for q in 1 2; do run cfg4 62 DS_CHAIN_PARTS=$q; done
done
tail -5 $O/err.txt
