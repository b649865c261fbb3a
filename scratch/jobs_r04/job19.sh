#!/bin/bash
# round 4, job 19: non-temporal tile traffic in the wide WPE kernel (W / taps ordinary), the operators' state buffers non-temporal, narrow WPE ordinary:
# whole GPU suite + the chains at one hop per call and in 10 s chunks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job19; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee -a $O/pytest_gpu.log
for cfg in wpe_nb cfg4_n10 cfg4 cfg5 cfg3; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wpe_nb T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
timeout 600 python bench.py --config cfg4 --steps 3 --warmup 1 --hops-per-step 312 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 T=312', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --hops-per-step 625 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 T=625', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
