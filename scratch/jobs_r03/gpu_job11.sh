#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
GPU_MAX_HW_QUEUES=8 bash scripts/profile_bench.sh r03e_cfg5 --config cfg5 --steps 20 > /dev/null 2>&1
for t in cfg5; do
  d=gpurun_out/prof_r03e_$t
  cp $d/traffic.json gpurun_out/r03e/${t}_traffic.json 2>/dev/null
  cp $d/kernel_stats.csv gpurun_out/r03e/${t}_kernel_stats.csv 2>/dev/null
  cp $d/summary.txt gpurun_out/r03e/${t}_summary.txt 2>/dev/null
  f=$(find $d/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r03e/${t}_rocprofv3_stats.csv
  rm -rf $d/trace $d/pmc_*/
done
python scripts/stage_budget.py cfg5 gpurun_out/r03e/cfg5_traffic.json > gpurun_out/r03e/cfg5_stage_budget.md 2> gpurun_out/r03e/cfg5_stage_budget.err
python scripts/stage_budget.py cfg4 gpurun_out/r03e/cfg4_traffic.json > gpurun_out/r03e/cfg4_stage_budget.md 2> gpurun_out/r03e/cfg4_stage_budget.err
grep "^|" gpurun_out/r03e/cfg5_stage_budget.md | cut -d'|' -f2,8,9,10
grep "^|" gpurun_out/r03e/cfg4_stage_budget.md | cut -d'|' -f2,8,9,10
python bench.py --steps 20 --warmup 5 > gpurun_out/r03e/bench_default_k20.json 2> gpurun_out/r03e/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03e/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'survey', r['frac_survey_bytes'], 'measured', r.get('frac_measured'))
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'], d['roofline_hbm'].get('frac_measured'))
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v['roofline']['bound'], v['roofline']['frac'], v['roofline'].get('frac_lds'), v['roofline'].get('frac_measured'))
PY
