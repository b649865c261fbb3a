#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
DS_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r03c/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03c/gpu_tests.txt 2>&1
tail -4 gpurun_out/r03c/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r03c/bench_default_k20.json 2> gpurun_out/r03c/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03c/bench_default_k20.json').read().strip().splitlines()[-1])
print('cfg2', d['value'], d['ms_per_step'], d['roofline']['frac'])
print('hbm', d['roofline_hbm']['value'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'])
PY
