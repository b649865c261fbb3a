#!/bin/bash
# final set of round 3, on the committed build: GPU tests with the parity log, rocprofv3 kernel trace + HBM PMC passes of the five one-hop workloads,
# SQ passes of the four 10 s-per-call workloads (-> compute_latest.json), stage budgets, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
bash scripts/profile_bench.sh r03k_cfg2 > /dev/null 2>&1
bash scripts/profile_bench.sh r03k_cfg2_hbm --config cfg2 --batch 16384 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh r03k_cfg3 --config cfg3 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh r03k_cfg4 --config cfg4 --steps 20 > /dev/null 2>&1
GPU_MAX_HW_QUEUES=8 bash scripts/profile_bench.sh r03k_cfg5 --config cfg5 --steps 20 > /dev/null 2>&1
for t in cfg2 cfg2_hbm cfg3 cfg4 cfg5; do
  d=gpurun_out/prof_r03k_$t
  cp $d/traffic.json $O/${t}_traffic.json 2>/dev/null; cp $d/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; cp $d/summary.txt $O/${t}_summary.txt 2>/dev/null
  f=$(find $d/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${t}_rocprofv3_stats.csv
  rm -rf $d/trace $d/pmc_*/
done
python scripts/stage_budget.py cfg5 $O/cfg5_traffic.json > $O/cfg5_stage_budget.md 2> $O/cfg5_stage_budget.err
python scripts/stage_budget.py cfg4 $O/cfg4_traffic.json > $O/cfg4_stage_budget.md 2> $O/cfg4_stage_budget.err
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03k_cfg2_T625 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03k_cfg3_T625 --config cfg3 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03k_cfg4_T312 --config cfg4 --hops-per-step 312 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03k_cfg5_T625 --config cfg5 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
python scripts/make_compute_latest.py cfg2_10s_chunks=gpurun_out/prof_r03k_cfg2_T625:640000 cfg3_10s_chunks=gpurun_out/prof_r03k_cfg3_T625:2560000 cfg4_10s_chunks=gpurun_out/prof_r03k_cfg4_T312:319488 cfg5_10s_chunks=gpurun_out/prof_r03k_cfg5_T625:1280000 > $O/compute_latest.json 2> $O/make_compute.err
for t in cfg2_T625 cfg3_T625 cfg4_T312 cfg5_T625; do cp gpurun_out/prof_r03k_$t/compute.json $O/${t}_compute.json 2>/dev/null; cp gpurun_out/prof_r03k_$t/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; rm -rf gpurun_out/prof_r03k_$t/trace gpurun_out/prof_r03k_$t/pmc_*/; done
mkdir -p profiles/r03k; cp $O/*_traffic.json $O/*_summary.txt profiles/r03k/
python scripts/make_traffic_latest.py profiles/r03k > $O/traffic_latest.json
cp $O/traffic_latest.json profiles/traffic_latest.json; cp $O/compute_latest.json profiles/compute_latest.json
python bench.py --steps 20 --warmup 5 > $O/bench_default_k20.json 2> $O/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03k/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'survey', r['frac_survey_bytes'], 'measured', r.get('frac_measured'))
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'], d['roofline_hbm'].get('frac_measured'))
for k,v in d['other_configs'].items(): print(k, v['value'], v['ms_per_step'], v['roofline']['bound'], v['roofline']['frac'], v['roofline'].get('frac_lds'), v['roofline'].get('frac_measured'))
print(json.dumps(d.get('latency'))[:600])
PY
