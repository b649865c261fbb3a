#!/bin/bash
# round 5: the driver's bench command once more on another box (box-to-box spread of the last build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05bx; mkdir -p $O
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/bench_detail_$1.json
python bench.py --steps 20 --warmup 5 > $O/bench_default_k20_$1.json 2> $O/bench_$1.err
python - $O/bench_default_k20_$1.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('cfg2', d['value'], d['roofline']['frac'], 'hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'])
print({k:(v['value'], v.get('frac')) for k,v in d['other_configs'].items()})
PY
