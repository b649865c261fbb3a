#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O
for v in work v2h v2n; do
  if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
  python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mcspp or notebook or steering or gev" 2>&1 | tail -1
done | tee $O/tests.txt
unset DSENH_LIB
for c in nb_mvdr nb_mvdr_m4; do
 for i in 1 2; do for v in work v2h v2n; do
  if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
  echo -n "$c $v  "
  timeout 120 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
 done; done
done 2>&1 | tee $O/ab.txt
for v in work v2h v2n; do
  if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
  for c in nb_mvdr nb_mvdr_m4; do echo -n "T625 $c $v "; python bench.py --config $c --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
done 2>&1 | tee $O/ab_T625.txt
export DSENH_LIB=$PWD/scratch/variants/libdsenh_v2h.so
PROFILE_HBM=0 bash scripts/profile_bench.sh r06c_nb --config nb_mvdr > /dev/null 2>&1; cp gpurun_out/prof_r06c_nb/kernel_stats.csv $O/nb_mvdr_v2h_kernel_stats.csv; head -8 $O/nb_mvdr_v2h_kernel_stats.csv
PROFILE_HBM=0 bash scripts/profile_bench.sh r06c_nb4 --config nb_mvdr_m4 > /dev/null 2>&1; cp gpurun_out/prof_r06c_nb4/kernel_stats.csv $O/nb_mvdr_m4_v2h_kernel_stats.csv; head -8 $O/nb_mvdr_m4_v2h_kernel_stats.csv
rm -rf gpurun_out/prof_r06c_nb/trace gpurun_out/prof_r06c_nb4/trace
