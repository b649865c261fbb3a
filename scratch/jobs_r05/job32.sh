#!/bin/bash
# round 5: the wide-tap WPE unit (ds_kernels_wpe.o) without the SLP vectoriser: Wpe.update 4 x 20 at one hop and with 10 s per call, cfg4_n10
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05nw; mkdir -p $O
for r in "T1 --steps 20 --warmup 5" "T2500 --steps 2 --warmup 1 --hops-per-step 2500"; do set -- $r; n=$1; shift; for i in 1 2; do for v in head5 nswpe; do
  echo -n "wpe_nb_$n $v  "; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 900 python bench.py --config wpe_nb "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/wpe_wide_noslp_ab.txt
