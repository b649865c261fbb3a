#!/bin/bash
# A/B: new4 = before the WPE work; splitasm = WPE with compile-time shapes + packed downdate; fusedasm = the same + every packed helper as one asm statement
cd $GRAFT_REPO_ROOT
run() { # variant config extra...
  v=$1; c=$2; shift 2
  DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 900 python bench.py --config $c --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('%-9s %-5s %s -> %.4g frames/s  %.5f ms/step' % ('$v', '$c', '$*', d['value'], d['ms_per_step']))
"
}
echo "== device check of the packed helpers (fused build in tree? no: HEAD = split)"; timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "packed_complex or wpe" 2>&1 | tail -2
for r in 1 2; do
  for v in new4 splitasm fusedasm; do
    run $v cfg4 --steps 20 --warmup 5
    run $v cfg4 --hops-per-step 312 --steps 2 --warmup 1
  done
  for v in splitasm fusedasm; do
    run $v cfg2 --steps 625 --warmup 25
    run $v cfg2 --hops-per-step 625 --steps 2 --warmup 1
    run $v cfg3 --steps 200 --warmup 25
    run $v cfg3 --hops-per-step 625 --steps 2 --warmup 1
    run $v cfg5 --steps 100 --warmup 10
  done
done
