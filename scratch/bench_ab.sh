#!/bin/bash
# Run ON THE GPU BOX: bench.py under each named variant library, alternating, twice.  Usage: bash scratch/bench_ab.sh v1 v2 [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
V1=$1; V2=$2; shift 2
for r in 1 2 3; do
  for V in $V1 $V2; do
    DSENH_LIB=$ROOT/scratch/variants/libdsenh_$V.so python $ROOT/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V', d['value'], d['roofline']['launch_ms'], d['roofline']['frac'])"
  done
done
