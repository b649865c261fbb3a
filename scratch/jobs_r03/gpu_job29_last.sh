#!/bin/bash
# last check of the committed tree: the GPU suite with its parity log, smoke(), the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > $O/bench_default_k20.json 2> $O/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03l/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'measured', r.get('frac_measured'), r['traffic_source'][:40])
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'])
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v['roofline']['bound'], v['roofline']['frac'])
PY
