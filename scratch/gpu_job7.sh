#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
DS_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r03e/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03e/gpu_tests.txt 2>&1
tail -4 gpurun_out/r03e/gpu_tests.txt
for c in cfg5 cfg4; do python bench.py --config $c --steps 40 --warmup 4 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_state_bytes'))"; done
bash scripts/profile_bench.sh r03e_cfg5 --config cfg5 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh r03e_cfg4 --config cfg4 --steps 20 > /dev/null 2>&1
for t in cfg5 cfg4; do rm -rf gpurun_out/prof_r03e_$t/trace gpurun_out/prof_r03e_$t/pmc_*/; grep -A12 "HBM traffic" gpurun_out/prof_r03e_$t/summary.txt | cut -c1-150; done
