#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
DS_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r03d/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03d/gpu_tests.txt 2>&1
tail -4 gpurun_out/r03d/gpu_tests.txt
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03d_cfg2_T625 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03d_cfg3_T625 --config cfg3 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03d_cfg4_T312 --config cfg4 --hops-per-step 312 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03d_cfg5_T625 --config cfg5 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
python scripts/make_compute_latest.py cfg2_10s_chunks=gpurun_out/prof_r03d_cfg2_T625:640000 cfg3_10s_chunks=gpurun_out/prof_r03d_cfg3_T625:2560000 cfg4_10s_chunks=gpurun_out/prof_r03d_cfg4_T312:319488 cfg5_10s_chunks=gpurun_out/prof_r03d_cfg5_T625:1280000 > profiles/compute_latest.json 2> gpurun_out/r03d/make_compute.err
cp profiles/compute_latest.json gpurun_out/r03d/
for t in cfg2_T625 cfg3_T625 cfg4_T312 cfg5_T625; do rm -rf gpurun_out/prof_r03d_$t/trace gpurun_out/prof_r03d_$t/pmc_*/; done
python bench.py --steps 20 --warmup 5 > gpurun_out/r03d/bench_default_k20.json 2> gpurun_out/r03d/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03d/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'survey', r['frac_survey_bytes'], 'measured', r.get('frac_measured'))
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'], d['roofline_hbm'].get('frac_measured'))
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v['roofline']['bound'], v['roofline']['frac'], v['roofline'].get('frac_lds'), v['roofline'].get('frac_measured'))
print(json.dumps(d.get('latency'))[:900])
PY
