#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/j9
echo "== full gpu suite (float4 state planes, dword stores)"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for r in 1 2; do
 for v in oldst new4; do
  for c in cfg5 cfg4; do
   echo "== $v $c run $r"
   DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config $c --steps 625 --warmup 25 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))
"
  done
 done
done
