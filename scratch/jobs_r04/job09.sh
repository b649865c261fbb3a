#!/bin/bash
# round 4, job 9: the whole GPU suite + wide WPE bench after the clean-up (two serial chunks, four partial sums in P x)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job09; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
for cfg in wpe_nb cfg4_n10; do
  export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_${cfg}.json
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wpe_nb T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
