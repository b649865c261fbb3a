#!/bin/bash
# round 5: the MVDR + post-filter kernel's second build for long calls (no SLP vectoriser, conjugation folded into the MVDR sweep): PF tests, call-length sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05pf; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -k "pf or postfilter or mvdr_pf or G23" 2>&1 | tail -3
for T in 1 2 4 8 16 64 625; do for i in 1 2; do for v in pflong_never pflong2; do
  [ $T = 1 ] && A="--steps 20 --warmup 5" || A="--steps 4 --warmup 2 --hops-per-step $T"
  echo -n "pf_T$T $v  "; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config mvdr_pf $A --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/pf_long_sweep.txt
