#!/bin/bash
# A/B: cur = committed build (+ T_run condition); gsc1 = address hoisting in the transform stages of the GSC kernel (128 VGPRs, no scratch since the state goes back inside the last hop)
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
  for v in cur gsc1; do
    run $v cfg3 --steps 200 --warmup 25
    run $v cfg3 --hops-per-step 625 --steps 2 --warmup 1
    run $v cfg3 --batch 1024 --steps 400 --warmup 25
  done
done
