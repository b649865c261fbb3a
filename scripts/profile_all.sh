#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the whole committed profile set of a round in one go — per bench workload the kernel trace and the two HBM PMC
# passes at one hop per call, the SQ passes + the HBM passes with 10 s per call — condensed into gpurun_out/<tag>/ (copy into profiles/<tag>/),
# and the two tables bench.py quotes (compute_latest.json, traffic_latest.json).
# Usage: bash scripts/profile_all.sh <tag> ["cfg2 cfg3 ..." (default: every workload of the bench line)]
set -u
R=${1:-rXX}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
O=gpurun_out/$R; mkdir -p $O
T1=${2:-"cfg2 cfg2_hbm mvdr_pf cfg3 cfg4 cfg5 wpe_nb cfg4_n10 nb_mvdr nb_mvdr_m4 tdgsc fdgsc"}
declare -A CH=( [cfg2]=625 [mvdr_pf]=625 [cfg3]=625 [cfg4]=312 [cfg5]=625 [wpe_nb]=2500 [nb_mvdr]=625 [nb_mvdr_m4]=625 )
declare -A FR=( [cfg2]=640000 [mvdr_pf]=640000 [cfg3]=2560000 [cfg4]=319488 [cfg5]=1280000 [wpe_nb]=2560000 [nb_mvdr]=1280000 [nb_mvdr_m4]=1280000 )
keep() {   # prof dir, key: the condensed files only
  d=$1; k=$2
  for f in traffic.json kernel_stats.csv summary.txt compute.json bench_line.json; do [ -f $d/$f ] && cp $d/$f $O/${k}_$f; done
  f=$(find $d/trace -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/${k}_rocprofv3_stats.csv
  for l in pmc_fetch pmc_sq; do [ -f $d/$l.log ] && grep '^{' $d/$l.log | tail -1 > $O/${k}_${l}_line.json; done
  rm -rf $d/trace $d/pmc_*/
}
for c in $T1; do
  case $c in
    cfg2_hbm) args="--config cfg2 --batch 16384 --steps 20";;
    cfg4_n10) args="--config cfg4_n10 --steps 10";;
    *) args="--config $c --steps 20";;
  esac
  bash scripts/profile_bench.sh ${R}_$c $args > /dev/null 2>&1
  keep gpurun_out/prof_${R}_$c $c
done
specs=""
for c in $T1; do
  T=${CH[$c]:-}; [ -z "$T" ] && continue
  PROFILE_SQ=1 bash scripts/profile_bench.sh ${R}_${c}_T$T --config $c --hops-per-step $T --steps 2 --warmup 1 > /dev/null 2>&1
  # how the chain's kernels share the GPU in time (the timeline that says why the stream pipeline stops where it does)
  python scripts/concurrency.py gpurun_out/prof_${R}_${c}_T$T > $O/${c}_T${T}_concurrency.txt 2>&1
  keep gpurun_out/prof_${R}_${c}_T$T ${c}_T$T
  specs="$specs ${c}_10s_chunks=$O/${c}_T$T:${FR[$c]}"            # the condensed copies: the table then names files that are committed
done
python scripts/make_compute_latest.py $specs > $O/compute_latest.json 2> $O/make_compute.err
python scripts/make_traffic_latest.py $O > $O/traffic_latest.json 2> $O/make_traffic.err
for c in cfg4 cfg5; do [ -f $O/${c}_traffic.json ] && python scripts/stage_budget.py $c $O/${c}_traffic.json > $O/${c}_stage_budget.md 2>> $O/make_traffic.err; done
ls $O | head -100; tail -3 $O/make_compute.err $O/make_traffic.err
