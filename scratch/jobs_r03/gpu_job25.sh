#!/bin/bash
# cfg4: utterance groups again with the round-3 WPE kernel (DS_CHAIN_PARTS = 1, 2 (default), 3, 4), interleaved
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift
  timeout 900 python bench.py --config cfg4 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%-9s cfg4 %s -> %.4g frames/s  %.5f ms/step' % ('$tag', '$*', d['value'], d['ms_per_step']))
"
}
for r in 1 2 3; do
  for n in 2 1 3 4; do DS_CHAIN_PARTS=$n run parts$n --steps 40 --warmup 5; done
done
for n in 2 3 4; do DS_CHAIN_PARTS=$n run parts$n --hops-per-step 312 --steps 2 --warmup 1; done
