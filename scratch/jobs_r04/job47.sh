#!/bin/bash
# round 4, job 47: the committed library (s_nop pad in front of the asm group store): operator / chain tests, cfg4 / cfg5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job47; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee -a $O/pytest.log
for cfg in cfg4 cfg5; do
  timeout 600 python bench.py --config $cfg --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
