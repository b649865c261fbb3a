#!/bin/bash
# round 4, job 13: wide WPE as one workgroup per bin, two lanes per row (ds_wpe_wide.hpp) against the one-wavefront form (job 8: 680 k / 149 k / 3.53 M)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job13; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_wide.log
for cfg in wpe_nb cfg4_n10 cfg4; do
  export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_${cfg}.json
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wpe_nb T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
DS_WPE_GENERIC=1 timeout 600 python bench.py --config wpe_nb --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('generic wpe_nb T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
