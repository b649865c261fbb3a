#!/bin/bash
O=gpurun_out/r06k; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mcspp or notebook or steering or gev" 2>&1 | tail -1
for c in nb_mvdr nb_mvdr_m4; do for b in 1024 992 512 496; do
  echo -n "$c B=$b "; python bench.py --config $c --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done 2>&1 | tee $O/batch_tail.txt
for c in nb_mvdr nb_mvdr_m4; do echo -n "T625 $c "; python bench.py --config $c --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee $O/T625.txt
