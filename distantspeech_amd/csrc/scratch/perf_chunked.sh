#!/bin/bash
# chain configs at one hop per call and chunked
for a in "cfg4" "cfg4 --hops-per-step 39" "cfg4 --hops-per-step 312" "cfg5" "cfg5 --hops-per-step 62" "cfg5 --hops-per-step 625"; do
  echo -n "$a  "
  timeout 280 python bench.py --config $a --steps 2 --warmup 1 --min-region-ms 100 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
