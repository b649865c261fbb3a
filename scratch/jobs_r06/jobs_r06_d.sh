#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt
for c in nb_mvdr nb_mvdr_m4; do for b in 1024 1020 1008 2040; do
  echo -n "$c B=$b "; python bench.py --config $c --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done 2>&1 | tee $O/batch_sweep.txt
for c in nb_mvdr nb_mvdr_m4; do echo -n "T625 $c "; python bench.py --config $c --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee $O/T625.txt
PROFILE_HBM=0 bash scripts/profile_bench.sh r06d_nb --config nb_mvdr > /dev/null 2>&1; cp gpurun_out/prof_r06d_nb/kernel_stats.csv $O/nb_mvdr_kernel_stats.csv; head -6 $O/nb_mvdr_kernel_stats.csv
PROFILE_HBM=0 bash scripts/profile_bench.sh r06d_nb4 --config nb_mvdr_m4 > /dev/null 2>&1; cp gpurun_out/prof_r06d_nb4/kernel_stats.csv $O/nb_mvdr_m4_kernel_stats.csv; head -6 $O/nb_mvdr_m4_kernel_stats.csv
rm -rf gpurun_out/prof_r06d_nb/trace gpurun_out/prof_r06d_nb4/trace
