#!/bin/bash
# A/B: notebook operator at 6 microphones, one wave per SIMD (348 registers) against two (256 + 276 B scratch); kernel trace of the default
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mcspp or notebook or steering or gev" > $O/tests_mcspp.txt 2>&1; tail -2 $O/tests_mcspp.txt
for c in nb_mvdr; do
  bash scratch/perf_ab_lib.sh w2 $c 2>&1 | tee $O/ab_w2_$c.txt
done
for v in work w2; do
  if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
  echo -n "T625 $v "; python bench.py --config nb_mvdr --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/ab_w2_T625.txt
unset DSENH_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o nb -- python3 $GRAFT_REPO_ROOT/bench.py --config nb_mvdr --steps 20 --warmup 5 --no-extras --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp $f $O/nb_mvdr_kernel_stats.csv; head -8 $O/nb_mvdr_kernel_stats.csv; rm -rf $O/trace
