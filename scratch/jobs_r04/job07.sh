#!/bin/bash
# round 4, job 7: first profile of the wide-tap WPE at the notebook's operating point (one hop per call; 250 hops per call)
cd $GRAFT_REPO_ROOT
PROFILE_SQ=1 bash scripts/profile_bench.sh r04a_wpe_nb --config wpe_nb --steps 20 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r04a_wpe_nb_T250 --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 > /dev/null 2>&1
cat gpurun_out/prof_r04a_wpe_nb/summary.txt | head -60
cat gpurun_out/prof_r04a_wpe_nb_T250/summary.txt | head -40
