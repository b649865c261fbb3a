#!/bin/bash
# DS_CHAIN_PRIO=<mask> (1 McSpp's stream, 2 the tail's, 4 the front end's, 8 the blocking filters') against the default, interleaved
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
for r in 1 2 3; do
  run base --steps 100 --warmup 10
  for m in 2 6 10 3 14; do DS_CHAIN_PRIO=$m run mask$m --steps 100 --warmup 10; done
done
for r in 1 2 3; do
  run base --hops-per-step 625 --steps 2 --warmup 1
  DS_CHAIN_PRIO=2 run mask2 --hops-per-step 625 --steps 2 --warmup 1
done
