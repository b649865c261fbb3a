#!/bin/bash
# round 5: the notebook online MVDR as one handle (tests, bench entries, kernel trace), the RCCL world-1 test, the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "online_mvdr or notebook" > $O/gpu_tests_nb.txt 2>&1; tail -15 $O/gpu_tests_nb.txt
timeout 600 python -m pytest tests/test_gpu_bench.py -m gpu -q -x > $O/gpu_tests_bench.txt 2>&1; tail -15 $O/gpu_tests_bench.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
for c in nb_mvdr nb_mvdr_m4; do
  for a in "--steps 20 --warmup 5" "--steps 2 --warmup 1 --hops-per-step 625"; do
    echo -n "$c $a  "; timeout 300 python bench.py --config $c $a --no-cpu-baseline --no-extras 2>$O/err_$c.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['bytes_per_launch'])"
  done
done
bash scripts/profile_bench.sh r05c_nb_mvdr --config nb_mvdr > $O/prof_nb_mvdr.txt 2>&1; head -12 $O/prof_nb_mvdr.txt; grep -A12 "HBM traffic" $O/prof_nb_mvdr.txt
PROFILE_HBM=0 bash scripts/profile_bench.sh r05c_nb_mvdr_T625 --config nb_mvdr --hops-per-step 625 --steps 2 --warmup 1 > $O/prof_nb_mvdr_T625.txt 2>&1; head -10 $O/prof_nb_mvdr_T625.txt
