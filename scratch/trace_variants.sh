#!/bin/bash
# Run ON THE GPU BOX: kernel-trace the cfg5 bench under each named variant library.  Usage: bash scratch/trace_variants.sh v1 v2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for V in "$@"; do
  export DSENH_LIB=$ROOT/scratch/variants/libdsenh_$V.so
  echo "== $V"
  bash $ROOT/scripts/trace_chain.sh cfg5_$V scripts/bench_cfg5.py 2>&1 | tail -14
done
