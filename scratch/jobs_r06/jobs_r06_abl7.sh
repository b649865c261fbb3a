#!/bin/bash
# nb_mvdr / nb_mvdr_m4, one block per call: the chain without its analysis + McCDR launch, without its synthesis (stale real data): what a
# stage pipeline over the calls could save at most
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
run() { c=$1; s=$2; DS_ABL_AFTER=40 DS_ABL_SKIP=$s python3 $R/bench.py --config $c --graph 0 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-10s skip=%-16s %7.3f M frames/s  %8.2f us per step' % ('$c', '$s', d['value']/1e6, d['ms_per_step']*1e3))" | tee -a $O/nb_mvdr_stage_ablation.txt; }
for c in nb_mvdr nb_mvdr_m4; do for s in none nbcdr nbistft "nbcdr,nbistft"; do run $c $s; done; done
