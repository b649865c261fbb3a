#!/bin/bash
# A/B: wpe2 = shipped; nyq3 = the Nyquist bin's pass of the per-bin program on the first lane of wave 3 instead of wave 0 (512-point frames)
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
  for v in wpe2 nyq3; do
    run $v cfg2 --steps 625 --warmup 25
    run $v cfg2 --hops-per-step 625 --steps 2 --warmup 1
    run $v cfg3 --steps 200 --warmup 25
    run $v cfg3 --hops-per-step 625 --steps 2 --warmup 1
    run $v fixed --hops-per-step 625 --steps 2 --warmup 1
    run $v cfg5 --steps 100 --warmup 10
    run $v cfg5 --hops-per-step 625 --steps 2 --warmup 1
  done
done
