#!/bin/bash
# round 4, profile set r04l on the last build of round 4 (state on 128-byte lines, streamed planes, one 16-byte store per operator state group): rocprofv3 kernel trace + HBM PMC passes of the one-hop workloads (cfg2, cfg2 at 16 384, cfg3,
# cfg4, cfg5, the wide-tap WPE at the notebook's operating point, cfg4 with 10 taps), SQ passes of the 10 s-per-call workloads
# (-> compute_latest.json), stage budgets
cd $GRAFT_REPO_ROOT
R=r04l; O=gpurun_out/$R; mkdir -p $O
bash scripts/profile_bench.sh ${R}_cfg2 > /dev/null 2>&1
bash scripts/profile_bench.sh ${R}_cfg2_hbm --config cfg2 --batch 16384 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh ${R}_cfg3 --config cfg3 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh ${R}_cfg4 --config cfg4 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh ${R}_cfg5 --config cfg5 --steps 20 > /dev/null 2>&1
PROFILE_SQ=1 bash scripts/profile_bench.sh ${R}_wpe_nb --config wpe_nb --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh ${R}_cfg4_n10 --config cfg4_n10 --steps 10 > /dev/null 2>&1
for t in cfg2 cfg2_hbm cfg3 cfg4 cfg5 wpe_nb cfg4_n10; do
  d=gpurun_out/prof_${R}_$t
  cp $d/traffic.json $O/${t}_traffic.json 2>/dev/null; cp $d/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; cp $d/summary.txt $O/${t}_summary.txt 2>/dev/null
  cp $d/compute.json $O/${t}_compute.json 2>/dev/null
  f=$(find $d/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${t}_rocprofv3_stats.csv
  rm -rf $d/trace $d/pmc_*/
done
python scripts/stage_budget.py cfg5 $O/cfg5_traffic.json > $O/cfg5_stage_budget.md 2> $O/cfg5_stage_budget.err
python scripts/stage_budget.py cfg4 $O/cfg4_traffic.json > $O/cfg4_stage_budget.md 2> $O/cfg4_stage_budget.err
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh ${R}_cfg2_T625 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh ${R}_cfg3_T625 --config cfg3 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh ${R}_cfg4_T312 --config cfg4 --hops-per-step 312 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh ${R}_cfg5_T625 --config cfg5 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh ${R}_wpe_nb_T2500 --config wpe_nb --hops-per-step 2500 --steps 2 --warmup 1 > /dev/null 2>&1
python scripts/make_compute_latest.py cfg2_10s_chunks=gpurun_out/prof_${R}_cfg2_T625:640000 cfg3_10s_chunks=gpurun_out/prof_${R}_cfg3_T625:2560000 cfg4_10s_chunks=gpurun_out/prof_${R}_cfg4_T312:319488 cfg5_10s_chunks=gpurun_out/prof_${R}_cfg5_T625:1280000 wpe_nb_10s_chunks=gpurun_out/prof_${R}_wpe_nb_T2500:2560000 > $O/compute_latest.json 2> $O/make_compute.err
for t in cfg2_T625 cfg3_T625 cfg4_T312 cfg5_T625 wpe_nb_T2500; do cp gpurun_out/prof_${R}_$t/compute.json $O/${t}_compute.json 2>/dev/null; cp gpurun_out/prof_${R}_$t/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof_${R}_$t/summary.txt $O/${t}_summary.txt 2>/dev/null; rm -rf gpurun_out/prof_${R}_$t/trace gpurun_out/prof_${R}_$t/pmc_*/; done
mkdir -p /tmp/tl; cp $O/*_traffic.json $O/*_summary.txt /tmp/tl/
python scripts/make_traffic_latest.py $O > $O/traffic_latest.json 2> $O/make_traffic.err
ls $O | head -80; tail -3 $O/make_compute.err $O/make_traffic.err
