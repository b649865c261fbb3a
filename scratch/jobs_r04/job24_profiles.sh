#!/bin/bash
# round 4, profile set r04d (continued): cfg3 after the GSC bin program gained its optional power export (static instruction count moved)
cd $GRAFT_REPO_ROOT
R=r04d; O=gpurun_out/$R; mkdir -p $O
bash scripts/profile_bench.sh ${R}_cfg3 --config cfg3 --steps 20 > /dev/null 2>&1
for t in cfg3; do
  d=gpurun_out/prof_${R}_$t
  cp $d/traffic.json $O/${t}_traffic.json 2>/dev/null; cp $d/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; cp $d/summary.txt $O/${t}_summary.txt 2>/dev/null
  f=$(find $d/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${t}_rocprofv3_stats.csv
  rm -rf $d/trace $d/pmc_*/
done
PROFILE_SQ=1 PROFILE_HBM=0 bash scripts/profile_bench.sh ${R}_cfg3_T625 --config cfg3 --hops-per-step 625 --steps 2 --warmup 1 > /dev/null 2>&1
python scripts/make_compute_latest.py cfg3_10s_chunks=gpurun_out/prof_${R}_cfg3_T625:2560000 > $O/compute_cfg3.json 2> $O/make_compute_cfg3.err
for t in cfg3_T625; do cp gpurun_out/prof_${R}_$t/compute.json $O/${t}_compute.json 2>/dev/null; cp gpurun_out/prof_${R}_$t/kernel_stats.csv $O/${t}_kernel_stats.csv 2>/dev/null; cp gpurun_out/prof_${R}_$t/summary.txt $O/${t}_summary.txt 2>/dev/null; rm -rf gpurun_out/prof_${R}_$t/trace gpurun_out/prof_${R}_$t/pmc_*/; done
mkdir -p $O/only_cfg3; cp $O/cfg3_traffic.json $O/cfg3_summary.txt $O/only_cfg3/
python scripts/make_traffic_latest.py $O/only_cfg3 > $O/traffic_cfg3.json 2> $O/make_traffic_cfg3.err
ls $O; tail -3 $O/make_compute_cfg3.err $O/make_traffic_cfg3.err
