#!/bin/bash
# A/B of two builds on one box: default library against scratch/variants/libdsenh_$1.so, config $2 (default cfg5)
V=$1; C=${2:-cfg5}
for i in 1 2 3; do for v in work $V; do
  if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
  echo -n "$C $v  "
  timeout 120 python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
