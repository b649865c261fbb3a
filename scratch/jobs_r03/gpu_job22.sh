#!/bin/bash
# A/B: prev = a counter kernel (ds_tick3_kernel) at the end of every group's step of the cfg4 chain; tick = all three counter advances ride in the synthesis launch
cd $GRAFT_REPO_ROOT
run() { v=$1; c=$2; shift 2
  DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 900 python bench.py --config $c --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%-9s %-5s %s -> %.4g frames/s  %.5f ms/step' % ('$v', '$c', '$*', d['value'], d['ms_per_step']))
"
}
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do
  for v in prev tick; do
    run $v cfg4 --steps 40 --warmup 5
    run $v cfg4 --hops-per-step 312 --steps 2 --warmup 1
  done
done
