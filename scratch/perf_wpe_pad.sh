#!/bin/bash
# WPE occupancy cap (dynamic LDS padding) against the cfg4 chain with 1 and 2 utterance groups
for pad in 0 7000 13500 19500 27000 40000; do for parts in 1 2; do
  echo -n "pad=$pad parts=$parts  "
  DS_WPE_LDS_PAD=$pad DS_CHAIN_PARTS=$parts timeout 120 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
