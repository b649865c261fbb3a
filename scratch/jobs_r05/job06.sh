#!/bin/bash
# round 5: the RLS blocking filters inside McSpp's launch (OP_MCSPP_STEADY_FAN): chain variant tests, cfg5 A/B against DS_CHAIN_FAN_SEPARATE=1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -q -x -k "subband or chain or cfg5 or gsc" > $O/gpu_tests_chain.txt 2>&1; tail -6 $O/gpu_tests_chain.txt
ab() {
  for i in 1 2 3; do for v in 0 1; do
    echo -n "$1 fan_separate=$v  "
    DS_CHAIN_FAN_SEPARATE=$v timeout 300 python bench.py --config cfg5 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['bytes_per_launch'])"
  done; done
}
( ab T1 "--steps 20 --warmup 5"; ab T625 "--steps 2 --warmup 1 --hops-per-step 625" ) > $O/cfg5_fan_in_mcspp_ab.txt 2>&1
cat $O/cfg5_fan_in_mcspp_ab.txt
