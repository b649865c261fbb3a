#!/bin/bash
# round 5, check of the committed tree: the GPU suite with its parity log, smoke(), the whole profile set, the driver's bench command
cd $GRAFT_REPO_ROOT
R=${1:-r05j}; O=gpurun_out/$R; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 scratch/micro/valu_rate > $O/valu_rate.txt 2>&1
timeout 3000 bash scripts/profile_all.sh $R > $O/profile_all.log 2>&1; cat $O/make_compute.err $O/make_traffic.err | tail -4
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/bench_detail.json
# the tables the line quotes are this run's own passes
cp $O/compute_latest.json profiles/compute_latest.json; cp $O/traffic_latest.json profiles/traffic_latest.json
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default_k20.json 2> $O/bench_default_k20.err
wc -c $O/bench_default_k20.json; tail -4 $O/bench_default_k20.err
python - $O <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]+'/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'measured', r.get('frac_measured'), r.get('resident'), d.get('collective'))
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'], d['roofline_hbm'].get('frac_measured'))
for k,v in d['other_configs'].items(): print(k, v)
print(d.get('latency_us'), d.get('cpu_baseline'))
PY
