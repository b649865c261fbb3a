#!/bin/bash
# round 4, profile set r04d: the workloads whose kernel changed after r04b (wide-tap WPE with non-temporal tile traffic): wpe_nb, cfg4_n10
cd $GRAFT_REPO_ROOT
R=r04d; O=gpurun_out/$R; mkdir -p $O
PROFILE_SQ=1 bash scripts/profile_bench.sh ${R}_wpe_nb --config wpe_nb --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh ${R}_cfg4_n10 --config cfg4_n10 --steps 10 > /dev/null 2>&1
for t in wpe_nb cfg4_n10; do
  d=gpurun_out/prof_${R}_$t
  cp $d/traffic.json $O/${t}_traffic.json 2>/dev/null; cp $d/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; cp $d/summary.txt $O/${t}_summary.txt 2>/dev/null
  cp $d/compute.json $O/${t}_compute.json 2>/dev/null
  f=$(find $d/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${t}_rocprofv3_stats.csv
  rm -rf $d/trace $d/pmc_*/
done
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh ${R}_wpe_nb_T2500 --config wpe_nb --hops-per-step 2500 --steps 2 --warmup 1 > /dev/null 2>&1
python scripts/make_compute_latest.py wpe_nb_10s_chunks=gpurun_out/prof_${R}_wpe_nb_T2500:2560000 > $O/compute_wpe_nb.json 2> $O/make_compute.err
for t in wpe_nb_T2500; do cp gpurun_out/prof_${R}_$t/compute.json $O/${t}_compute.json 2>/dev/null; cp gpurun_out/prof_${R}_$t/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof_${R}_$t/summary.txt $O/${t}_summary.txt 2>/dev/null; rm -rf gpurun_out/prof_${R}_$t/trace gpurun_out/prof_${R}_$t/pmc_*/; done
python scripts/make_traffic_latest.py $O > $O/traffic_partial.json 2> $O/make_traffic.err
ls $O; tail -3 $O/make_compute.err $O/make_traffic.err
