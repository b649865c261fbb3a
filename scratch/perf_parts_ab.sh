#!/bin/bash
# A/B of the pipelined utterance groups of the cfg4 chain (DS_CHAIN_PARTS), same box; graph replay and plain launches
for g in 1 0; do for p in 1 2 4 1 2; do
  echo -n "graph=$g parts=$p  "
  DS_CHAIN_PARTS=$p timeout 120 python bench.py --config cfg4 --steps 20 --warmup 5 --graph $g --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
