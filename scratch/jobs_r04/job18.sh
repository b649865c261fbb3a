#!/bin/bash
# round 4, job 18: cfg4_n10 (wide WPE kernel inside the chain) and cfg4 per kernel, plain state stores against non-temporal ones
cd $GRAFT_REPO_ROOT
for lib in plain ntst ntld; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/libdsenh_$lib.so
  PROFILE_HBM=1 bash scripts/profile_bench.sh r04_j18_${lib}_cfg4_n10 --config cfg4_n10 > /dev/null 2>&1
  PROFILE_HBM=1 bash scripts/profile_bench.sh r04_j18_${lib}_cfg4 --config cfg4 > /dev/null 2>&1
  rm -rf gpurun_out/prof_r04_j18_${lib}_*/trace gpurun_out/prof_r04_j18_${lib}_*/pmc_*/
done
for lib in plain ntst ntld; do for c in cfg4_n10 cfg4; do echo "== $lib $c"; cat gpurun_out/prof_r04_j18_${lib}_$c/summary.txt; done; done
