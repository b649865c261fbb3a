#!/bin/bash
# cfg5 A/B on one box: fused front end (default) vs separate kernels; side stream vs one stream
cd "$(dirname "$0")/.."
for v in "" "DS_CHAIN_UNFUSED=1" "DS_CHAIN_NO_FORK=1" "DS_CHAIN_UNFUSED=1 DS_CHAIN_NO_FORK=1"; do
  for r in 1 2; do
    echo -n "[$v] "
    env $v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config cfg5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
