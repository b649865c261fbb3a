#!/bin/bash
# cfg5 pipeline toggles again, now that the tail's stream has priority: default against DS_CHAIN_NO_EARLY=1, DS_CHAIN_MAIN_JOIN=1, DS_CHAIN_PRIO=0, interleaved
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift
  timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%-12s cfg5 -> %.4g frames/s  %.5f ms/step' % ('$tag', d['value'], d['ms_per_step']))
"
}
for r in 1 2 3 4; do
  run default
  DS_CHAIN_NO_EARLY=1 run no_early
  DS_CHAIN_MAIN_JOIN=1 run main_join
  DS_CHAIN_PRIO=0 run prio0
done
