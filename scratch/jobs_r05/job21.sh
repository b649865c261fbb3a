#!/bin/bash
# round 5: the Jacobi eigen-solve's stopping threshold (|a_pq| <= 1e-17 max|a_ii| by default) at 1e-13 and 1e-11: parity of the notebook MVDR tests, nb_mvdr rate
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
for v in default jac26 jac22; do
  L=distantspeech_amd/libdsenh.so; [ $v != default ] && L=scratch/variants/libdsenh_$v.so
  export DSENH_LIB=$GRAFT_REPO_ROOT/$L
  DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_$v.jsonl timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "mcspp or online or gev or steering or linalg or notebook" 2>&1 | tail -2
  for i in 1 2; do echo -n "nb_mvdr $v "; timeout 300 python bench.py --config nb_mvdr --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
  echo -n "nb_mvdr_m4 $v "; timeout 300 python bench.py --config nb_mvdr_m4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee $O/jacobi_threshold_ab.txt
for v in default jac26 jac22; do echo == $v; grep -h "G11\|G19" $O/parity_$v.jsonl | cut -c1-330; done
