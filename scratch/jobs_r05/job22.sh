#!/bin/bash
# round 5: Jacobi rotation angles with one square root, one reciprocal and one reciprocal square root (hardware seed + two Newton steps) instead of three IEEE square roots and three divisions: parity of the notebook / GEV tests, nb_mvdr rates
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_jacfast.jsonl timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -m gpu -q -k "mcspp or online or gev or steering or linalg or notebook or pmwf or ban" 2>&1 | tail -2
grep -h "G11\|G19" $O/parity_jacfast.jsonl | cut -c1-400
for i in 1 2 3; do for v in mvdrc jacfast; do for c in nb_mvdr nb_mvdr_m4; do
  echo -n "$c $v "; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/jacobi_fast_ab.txt
for v in mvdrc jacfast; do echo -n "nb_mvdr_T625 $v "; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config nb_mvdr --steps 2 --warmup 1 --hops-per-step 625 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done 2>&1 | tee -a $O/jacobi_fast_ab.txt
