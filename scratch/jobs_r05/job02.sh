#!/bin/bash
# round 5: A/B of the one-pass MVDR + post-filter kernel's register budget on one box: default (4 waves per SIMD, no hoisting, lean McMcra)
# against 3 waves + hoisting level 2 (the first build), 3 waves without hoisting, and 4 waves with the products held across the inverse
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
run() { # name, extra bench args
  for v in work pf_w3h2 pf_w3h0 pf_nolean; do
    if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
    echo -n "$1 $v  "
    timeout 120 python bench.py --config mvdr_pf $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
}
( for i in 1 2 3; do run T1 "--steps 20 --warmup 5"; done
  for i in 1 2; do run T625 "--steps 2 --warmup 1 --hops-per-step 625"; done
  for i in 1 2; do run B16384 "--steps 20 --warmup 5 --batch 16384"; done ) > $O/pf_register_budget_ab.txt 2>&1
cat $O/pf_register_budget_ab.txt
unset DSENH_LIB
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "postfilter or adaptive" 2>&1 | tail -3
