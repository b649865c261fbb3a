#!/bin/bash
# A/B: splitasm = WPE with compile-time shapes; wpe2 = + two partial sums, the delayed-input prefetch without its scratch slot / flat load;
# gschoist1 = splitasm with address hoisting in the transform stages of the GSC kernel (44 B of scratch)
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
for r in 1 2; do
  for v in splitasm wpe2; do
    run $v cfg4 --steps 20 --warmup 5
    run $v cfg4 --hops-per-step 312 --steps 2 --warmup 1
  done
  for v in splitasm gschoist1; do
    run $v cfg3 --steps 200 --warmup 25
    run $v cfg3 --hops-per-step 625 --steps 2 --warmup 1
  done
done
