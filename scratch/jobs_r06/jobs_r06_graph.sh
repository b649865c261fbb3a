#!/bin/bash
# hipGraph replay against plain launches, one block per call, for the workloads bench.py replays as graphs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white
run() { c=$1; g=$2; python3 $R/bench.py --config $c --graph $g --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-10s graph=%s %8.3f M frames/s  %9.5f ms per step' % ('$c', '$g', d['value']/1e6, d['ms_per_step']))" | tee -a $O/graph_ab.txt; }
for rep in 1 2; do for c in cfg4 cfg3 nb_mvdr nb_mvdr_m4 wpe_nb mvdr_pf cfg2; do run $c 1; run $c 0; done; done
