#!/bin/bash
# round 5: fp64 issue costs (valu_rate), the whole profile set of the round (scripts/profile_all.sh), the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 300 scratch/micro/valu_rate > $O/valu_rate.txt 2>&1; grep -E "f64|v_fma_f32 \(3|v_pk_fma_f32 " $O/valu_rate.txt | grep "4 waves"
timeout 3000 bash scripts/profile_all.sh r05g > $O/profile_all.log 2>&1; tail -5 $O/profile_all.log
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/bench_detail.json
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default_k20.json 2> $O/bench_default_k20.err
wc -c $O/bench_default_k20.json; tail -4 $O/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05g/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'measured', r.get('frac_measured'), r.get('resident'))
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'], d['roofline_hbm'].get('frac_measured'))
for k,v in d['other_configs'].items(): print(k, v)
print(d.get('latency_us'), d.get('cpu_baseline'))
PY
