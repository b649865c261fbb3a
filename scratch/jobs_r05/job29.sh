#!/bin/bash
# round 5: the RLS fan kernel at five waves per SIMD (its error spectra through one buffer descriptor: 100 -> 84 registers): chain tests, cfg5 A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f5; mkdir -p $O
DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_fan5.so timeout 1200 python -m pytest tests -m gpu -q -k "subband or chain or cfg5 or gsc or subrls or fan" 2>&1 | tail -3
for r in "T1 --steps 20 --warmup 5" "T625 --steps 2 --warmup 1 --hops-per-step 625"; do set -- $r; n=$1; shift; for i in 1 2 3; do for v in head4 fan5; do
  echo -n "cfg5_$n $v  "; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config cfg5 "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/cfg5_fan5_ab.txt
