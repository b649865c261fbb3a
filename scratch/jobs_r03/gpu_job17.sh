#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wpe or stage_info or transform" 2>&1 | tail -3
python scripts/stage_budget.py cfg5 profiles/r03e/cfg5_traffic.json > gpurun_out/r03f/cfg5_stage_budget.md 2> gpurun_out/r03f/cfg5_stage_budget.err
python scripts/stage_budget.py cfg4 profiles/r03e/cfg4_traffic.json > gpurun_out/r03f/cfg4_stage_budget.md 2> gpurun_out/r03f/cfg4_stage_budget.err
grep "^|" gpurun_out/r03f/cfg5_stage_budget.md | cut -d'|' -f2,8,9,10
grep "^|" gpurun_out/r03f/cfg4_stage_budget.md | cut -d'|' -f2,8,9,10
tail -2 gpurun_out/r03f/*.err
