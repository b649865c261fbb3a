#!/bin/bash
# round 5: -fno-slp-vectorize per translation unit: nsa = GSC + MVDR/post-filter frame kernels, nsb = + the operator TU, nsc = + the chain tail's TU; head3 = none
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z3; mkdir -p $O
ab() {
  for i in 1 2; do for v in head3 nsa nsb nsc; do
    echo -n "$1 $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config $3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
}
( ab cfg3_T1 "--steps 20 --warmup 5" cfg3; ab pf_T1 "--steps 20 --warmup 5" mvdr_pf; ab cfg5_T1 "--steps 20 --warmup 5" cfg5; ab cfg5_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg5; ab cfg4_T1 "--steps 20 --warmup 5" cfg4; ab cfg4_T312 "--steps 2 --warmup 1 --hops-per-step 312" cfg4; ab nb_T1 "--steps 20 --warmup 5" nb_mvdr; ab wpe_nb_T1 "--steps 20 --warmup 5" wpe_nb ) > $O/noslp_tu_ab.txt 2>&1
cat $O/noslp_tu_ab.txt
