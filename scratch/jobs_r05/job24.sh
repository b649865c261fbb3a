#!/bin/bash
# round 5: the library built with -fno-slp-vectorize (the hand-packed complex helpers stay; the vectoriser's own pairings, which cost two moves per packed instruction in the real-symmetric McMcra code, go): every workload with 10 s per call, and cfg2 / cfg3 at one hop
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z2; mkdir -p $O
ab() {
  for i in 1 2; do for v in head3 noslp; do
    echo -n "$1 $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config $3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
}
( ab cfg3_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg3; ab cfg2_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg2; ab pf_T625 "--steps 2 --warmup 1 --hops-per-step 625" mvdr_pf; ab cfg5_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg5; ab cfg4_T312 "--steps 2 --warmup 1 --hops-per-step 312" cfg4; ab nb_T625 "--steps 2 --warmup 1 --hops-per-step 625" nb_mvdr; ab cfg3_T1 "--steps 20 --warmup 5" cfg3; ab cfg2_T1 "--steps 20 --warmup 5" cfg2 ) > $O/noslp_ab.txt 2>&1
cat $O/noslp_ab.txt
