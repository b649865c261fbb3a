#!/bin/bash
# notebook-MVDR chain as utterance groups: equality tests, then DS_CHAIN_PARTS = 1 / 2 / 3 on both workloads
O=gpurun_out/r06n; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -q -m gpu -k "mcspp or notebook or steering or gev" 2>&1 | tail -2
for c in nb_mvdr nb_mvdr_m4; do for i in 1 2; do for g in 1 2 3; do
  echo -n "$c parts=$g  "; DS_CHAIN_PARTS=$g python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/parts_ab.txt
for c in nb_mvdr nb_mvdr_m4; do for g in 1 2; do echo -n "T625 $c parts=$g "; DS_CHAIN_PARTS=$g python bench.py --config $c --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done 2>&1 | tee -a $O/parts_ab.txt
