#!/bin/bash
# experiment: DS_CHAIN_PRIO=<n>: one stream of the cfg5 chain at the greatest stream priority (1 McSpp's, 2 the tail's, 3 the front end's, 4 the blocking filters')
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift
  timeout 900 python bench.py --config cfg5 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%-9s cfg5 %s -> %.4g frames/s  %.5f ms/step' % ('$tag', '$*', d['value'], d['ms_per_step']))
"
}
for r in 1 2; do
  run base --steps 100 --warmup 10
  for n in 1 2 3 4; do DS_CHAIN_PRIO=$n run prio$n --steps 100 --warmup 10; done
done
