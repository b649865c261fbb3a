#!/bin/bash
# the notebook chain pipelined over the calls (default) against every call on the handle's stream (DS_NBMVDR_SERIAL=1): tests, then rates
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
(cd $R && python3 -m pytest tests -q -m gpu -x -k "notebook or mcspp_mvdr or nb_mvdr or OnlineMvdr or online_mvdr" 2>&1 | tail -4)
export DS_BENCH_SYNTH=white
run() { c=$1; t=$2; e=$3; shift 3; env $e python3 $R/bench.py --config $c --hops-per-step $t "$@" --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-10s T=%-4s %-20s %7.3f M frames/s  %8.2f us per step' % ('$c', '$t', '$e', d['value']/1e6, d['ms_per_step']*1e3))" | tee -a $O/nb_mvdr_pipeline_ab.txt; }
for rep in 1 2; do for c in nb_mvdr nb_mvdr_m4; do
run $c 1 X=0
run $c 1 DS_NBMVDR_SERIAL=1
run $c 4 X=0
run $c 4 DS_NBMVDR_SERIAL=1
done; done
run nb_mvdr 625 X=0 --steps 3 --warmup 1
run nb_mvdr 625 DS_NBMVDR_SERIAL=1 --steps 3 --warmup 1
