#!/bin/bash
# round 5: McSpp's steady-state operator (ds_binop_kernel<13,6>) pinned to 3 / 4 waves per SIMD (80 B / 332 B of scratch) against the 2-wave default: cfg5 with 10 s per call
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
ab() {
  for i in 1 2 3; do for v in default mcspp_w3 mcspp_w4; do
    echo -n "$1 $v  "
    L=distantspeech_amd/libdsenh.so; [ $v != default ] && L=scratch/variants/libdsenh_$v.so
    DSENH_LIB=$GRAFT_REPO_ROOT/$L timeout 300 python bench.py --config cfg5 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
}
( ab T625 "--steps 2 --warmup 1 --hops-per-step 625" ) > $O/cfg5_mcspp_waves_ab.txt 2>&1
cat $O/cfg5_mcspp_waves_ab.txt
