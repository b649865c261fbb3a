#!/bin/bash
# round 4, job 10: the SubbandGSC chain's front end as one kernel — parity of the variants, then cfg5 one hop per call / 10 s per call, fused vs three kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job10; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "subband or chain or cfg5 or SubbandGSC" 2>&1 | tail -6 | tee $O/pytest.log
for unf in 0 1; do
  export DS_CHAIN_FRONT_UNFUSED=$unf
  for rep in 1 2; do
  export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/detail_cfg5_unf$unf.json
  timeout 600 python bench.py --config cfg5 --steps 40 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FRONT_UNFUSED=$unf cfg5 T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 --hops-per-step 625 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FRONT_UNFUSED=$unf cfg5 T=625', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
