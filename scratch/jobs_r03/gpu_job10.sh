#!/bin/bash
# round-3 profile set (after the float4 operator-state planes): kernel trace + HBM PMC passes of the five BASELINE workloads at one hop per call,
# the per-stage byte budgets of the two chains
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
bash scripts/profile_bench.sh r03e_cfg2 > /dev/null 2>&1
bash scripts/profile_bench.sh r03e_cfg2_hbm --config cfg2 --batch 16384 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh r03e_cfg3 --config cfg3 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh r03e_cfg4 --config cfg4 --steps 20 > /dev/null 2>&1
GPU_MAX_HW_QUEUES=8 bash scripts/profile_bench.sh r03e_cfg5 --config cfg5 --steps 20 > /dev/null 2>&1
for t in cfg2 cfg2_hbm cfg3 cfg4 cfg5; do
  d=gpurun_out/prof_r03e_$t
  cp $d/traffic.json gpurun_out/r03e/${t}_traffic.json 2>/dev/null
  cp $d/kernel_stats.csv gpurun_out/r03e/${t}_kernel_stats.csv 2>/dev/null
  cp $d/summary.txt gpurun_out/r03e/${t}_summary.txt 2>/dev/null
  f=$(find $d/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r03e/${t}_rocprofv3_stats.csv
  rm -rf $d/trace $d/pmc_*/
done
python scripts/stage_budget.py cfg5 gpurun_out/r03e/cfg5_traffic.json > gpurun_out/r03e/cfg5_stage_budget.md 2> gpurun_out/r03e/cfg5_stage_budget.err
python scripts/stage_budget.py cfg4 gpurun_out/r03e/cfg4_traffic.json > gpurun_out/r03e/cfg4_stage_budget.md 2> gpurun_out/r03e/cfg4_stage_budget.err
cat gpurun_out/r03e/cfg5_stage_budget.md gpurun_out/r03e/cfg4_stage_budget.md | grep -v "^<!--"
tail -3 gpurun_out/r03e/*.err
