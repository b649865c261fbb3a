#!/bin/bash
# round 4, job 16: non-temporal state traffic in the RLS-WPE kernels (LDS-DMA with cache policy nt, store_state / load_state), wait as builtin.
# before: wpe_nb 680 k (0.645) / 3.53 M chunked, cfg4_n10 149 k (0.627), cfg4 1.605 M (0.65) / 7.06 M chunked
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job16; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1200 python -m pytest tests/test_gpu_wpe_wide.py tests/test_gpu_ops.py -x -q -m gpu -k "wpe or Wpe or WPE" 2>&1 | tail -3 | tee -a $O/pytest_wpe.log
for cfg in wpe_nb cfg4_n10 cfg4; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wpe_nb T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
timeout 600 python bench.py --config cfg4 --steps 3 --warmup 1 --hops-per-step 312 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 T=312', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
