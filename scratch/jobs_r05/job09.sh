#!/bin/bash
# round 5: the FIR bank with channel pairs in packed multiply-adds (ds_fir2_kernel) against one channel at a time (-DDS_FIR_SCALAR)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -q -x -k "fir or frontend or front or subband or tdgsc or fdgsc or time_alignment or TimeAlignment" > $O/gpu_tests_fir.txt 2>&1; tail -4 $O/gpu_tests_fir.txt
ab() {
  for i in 1 2 3; do for v in work fir_scalar; do
    if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
    echo -n "$1 $v  "
    timeout 300 python bench.py --config $3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done; done
}
( ab cfg5_T1 "--steps 20 --warmup 5" cfg5; ab cfg5_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg5; ab tdgsc_T1 "--steps 20 --warmup 5" tdgsc; ab tdgsc_T40 "--steps 4 --warmup 1 --hops-per-step 40" tdgsc ) > $O/fir_pairs_ab.txt 2>&1
cat $O/fir_pairs_ab.txt
unset DSENH_LIB
PROFILE_HBM=0 bash scripts/profile_bench.sh r05i_cfg5_T625 --config cfg5 --hops-per-step 625 --steps 2 --warmup 1 > $O/prof.txt 2>&1; head -12 $O/prof.txt
