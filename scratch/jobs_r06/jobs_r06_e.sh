#!/bin/bash
O=gpurun_out/r06e; mkdir -p $O
python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
for c in nb_mvdr nb_mvdr_m4; do
  echo -n "$c "; python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  echo -n "T625 $c "; python bench.py --config $c --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/nb.txt
