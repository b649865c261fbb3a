#!/bin/bash
# cfg5, 10 s per call: the front end's stream is the chain's critical path there (scripts/timeline_rows.py) — does a stream priority shorten it?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white
A="--config cfg5 --hops-per-step 625 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for rep in 1 2; do for m in 0 4 5 12 6 13; do
  DS_CHAIN_PRIO=$m python3 $R/bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prio mask=%-3s %6.2f M frames/s  %7.2f ms per 625-block call' % ('$m', d['value']/1e6, d['ms_per_step']))" | tee -a $O/cfg5_T625_prio.txt
done; done
