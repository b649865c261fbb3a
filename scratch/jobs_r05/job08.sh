#!/bin/bash
# round 5: the Jacobi eigen-solve on the upper triangle with converged rotations skipped: tests of everything that uses it, nb_mvdr rates
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "online_mvdr or notebook or mcspp or gev or steering or linalg or pmwf" > $O/gpu_tests_nb.txt 2>&1; tail -4 $O/gpu_tests_nb.txt
grep -h "G11\|G19" $O/parity_measured.jsonl | cut -c1-300
for c in nb_mvdr nb_mvdr_m4; do
  for a in "--steps 20 --warmup 5" "--steps 2 --warmup 1 --hops-per-step 625"; do
    echo -n "$c $a  "; timeout 300 python bench.py --config $c $a --no-cpu-baseline --no-extras 2>$O/err_$c.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  done
done
