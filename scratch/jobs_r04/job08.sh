#!/bin/bash
# round 4, job 8: wide WPE — four chunks through two tile buffers (pipelined) against two chunks one after the other
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job08; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_wide.log
for nch in 0 2; do
  export DS_WPE_WIDE_NCH=$nch
  for cfg in wpe_nb cfg4_n10; do
    export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_${cfg}_nch$nch.json
    timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NCH=$nch $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NCH=$nch wpe_nb T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
