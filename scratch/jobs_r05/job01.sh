#!/bin/bash
# round 5, first GPU run: the new one-pass MVDR + post-filter handle (tests first), then the whole GPU suite, smoke, the driver's bench
# command, and the trace + PMC passes of the mvdr_pf workload
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "postfilter" > $O/gpu_tests_pf.txt 2>&1; tail -5 $O/gpu_tests_pf.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/bench_detail.json
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default_k20.json 2> $O/bench_default_k20.err
wc -c $O/bench_default_k20.json; tail -4 $O/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05a/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'measured', r.get('frac_measured'), r.get('resident'))
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'], d['roofline_hbm'].get('frac_measured'))
for k,v in d['other_configs'].items(): print(k, v)
print(d.get('latency_us'), d.get('cpu_baseline'))
PY
unset DS_BENCH_DETAIL
bash scripts/profile_bench.sh r05a_mvdr_pf --config mvdr_pf > $O/prof_mvdr_pf.txt 2>&1; tail -15 $O/prof_mvdr_pf.txt
