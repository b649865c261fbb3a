#!/bin/bash
# round 5: mvdr_output without the conjugated copy of the covariance (823 -> 805 vector instructions in the headline kernel): A/B cfg2 / mvdr_pf / cfg4 in both regimes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
ab() {
  for i in 1 2 3; do for v in nomov mvdrc; do
    echo -n "$1 $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config $3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
}
( ab cfg2_T1 "--steps 20 --warmup 5" cfg2; ab cfg2_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg2; ab pf_T1 "--steps 20 --warmup 5" mvdr_pf; ab pf_T625 "--steps 2 --warmup 1 --hops-per-step 625" mvdr_pf; ab cfg4_T1 "--steps 20 --warmup 5" cfg4; ab cfg4_T312 "--steps 2 --warmup 1 --hops-per-step 312" cfg4 ) > $O/mvdr_noconj_ab.txt 2>&1
cat $O/mvdr_noconj_ab.txt
