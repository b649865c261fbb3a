#!/bin/bash
# A/B: non-temporal state accesses in the WPE and per-bin operator kernels (default build) against -DDS_PLAIN_STATE, same box
for c in cfg4 cfg5 cfg4 cfg5; do for v in nt plain; do
  if [ $v = plain ]; then export DSENH_LIB=$PWD/scratch/variants/libdsenh_plain.so; else unset DSENH_LIB; fi
  echo -n "$c $v  "
  timeout 120 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
