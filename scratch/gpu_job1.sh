#!/bin/bash
# round 3, GPU job 1: VALU issue micro-benchmark + SQ counters of every config in the 10 s-per-call regime
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
./scratch/micro/valu_rate > gpurun_out/r03a/valu_rate.txt 2>&1
(cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r03a/counters_avail.txt 2>&1)
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03a_cfg2_T625 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03a_cfg3_T625 --config cfg3 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03a_cfg4_T312 --config cfg4 --hops-per-step 312 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh r03a_cfg5_T625 --config cfg5 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03a/bench_default_k20.json 2> gpurun_out/r03a/bench_default_k20.err
for t in cfg2_T625 cfg3_T625 cfg4_T312 cfg5_T625; do rm -rf gpurun_out/prof_r03a_$t/trace gpurun_out/prof_r03a_$t/pmc_*/; done
tail -c 1500 gpurun_out/r03a/valu_rate.txt
