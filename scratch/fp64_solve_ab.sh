#!/bin/bash
# fp32 (default) against fp64 (-DDS_SOLVE_FP64) Hermitian solve in the fused frame kernels: parity against the golden fixtures and speed
mkdir -p gpurun_out
for v in fp32 fp64; do
  if [ $v = fp64 ]; then export DSENH_LIB=$PWD/scratch/variants/libdsenh_fp64.so; else unset DSENH_LIB; fi
  rm -f gpurun_out/parity_$v.jsonl
  DS_PARITY_LOG=$PWD/gpurun_out/parity_$v.jsonl timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -1
  for c in cfg2 "cfg2 --hops-per-step 125"; do
    echo -n "$v $c  "
    timeout 120 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['launch_ms'], d['roofline']['frac'])"
  done
  for shape in "8 1024" "8 512" "6 512"; do
    echo -n "$v M,nfft=$shape  "; timeout 100 python scratch/perf_shape_one.py --child $shape 2>/dev/null | tail -1
  done
done
python - <<'PY'
import json
a={};b={}
for n,d in (("fp32",a),("fp64",b)):
    for l in open("gpurun_out/parity_%s.jsonl"%n):
        r=json.loads(l); d[r.get("name")]=r
for k in a:
    if k in b:
        print(k, {kk:(a[k][kk], b[k].get(kk)) for kk in a[k] if kk not in ("name",) and isinstance(a[k][kk],(int,float))})
PY
