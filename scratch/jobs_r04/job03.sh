#!/bin/bash
# round 4, job 3: wide WPE with straight-line gathers and the four-instruction downdate
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job03; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_wide.log
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_wpe_nb.json
timeout 600 python bench.py --config wpe_nb --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_cfg4_n10.json
timeout 600 python bench.py --config cfg4_n10 --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4_n10 T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
cat $O/parity_measured.jsonl | grep stream
