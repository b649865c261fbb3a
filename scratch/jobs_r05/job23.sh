#!/bin/bash
# round 5: MCRA's indicator as a multiply-compare instead of an IEEE division + compare: GPU suite (parity log), A/B cfg2 / mvdr_pf / cfg3 in both regimes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y2; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
ab() {
  for i in 1 2 3; do for v in head2 mcradiv; do
    echo -n "$1 $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config $3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
}
( ab cfg2_T1 "--steps 20 --warmup 5" cfg2; ab cfg2_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg2; ab pf_T1 "--steps 20 --warmup 5" mvdr_pf; ab pf_T625 "--steps 2 --warmup 1 --hops-per-step 625" mvdr_pf ) > $O/mcra_div_ab.txt 2>&1
cat $O/mcra_div_ab.txt
