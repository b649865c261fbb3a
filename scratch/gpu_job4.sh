#!/bin/bash
cd $GRAFT_REPO_ROOT
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03b_cfg2_T625_pipe --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03b_fixed_T625_pipe --config fixed --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
DS_PIPE_MIN_T=100000000 PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03b_fixed_T625_frame --config fixed --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
for t in cfg2_T625_pipe fixed_T625_pipe fixed_T625_frame; do rm -rf gpurun_out/prof_r03b_$t/trace gpurun_out/prof_r03b_$t/pmc_*/; grep -v "^\"\|^{" gpurun_out/prof_r03b_$t/summary.txt | grep -v "distribution" | head -45; done
