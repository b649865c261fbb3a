#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
for i in 1 2 3; do for v in work m4w3; do
  if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
  echo -n "nb_mvdr_m4 $v  "; python bench.py --config nb_mvdr_m4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee $O/m4_waves_ab.txt
for v in work m4w3; do
  if [ $v = work ]; then unset DSENH_LIB; else export DSENH_LIB=$PWD/scratch/variants/libdsenh_$v.so; fi
  echo -n "T625 nb_mvdr_m4 $v "; python bench.py --config nb_mvdr_m4 --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee -a $O/m4_waves_ab.txt
unset DSENH_LIB
WPE_SAMPLE_ONLY_A=1 DS_PARITY_LOG=$PWD/$O/r06_wpe_sample_a2.jsonl python scratch/wpe_sample.py 32 2>&1 | tail -3 | cut -c1-300
