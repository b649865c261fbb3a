#!/bin/bash
# round 5: the notebook MVDR operator after its register diet (512 registers + 260 B of scratch -> 466 without scratch at 6 microphones): tests, rates, SQ counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "online_mvdr or notebook or mcspp or gev or steering" > $O/gpu_tests_nb.txt 2>&1; tail -5 $O/gpu_tests_nb.txt
for c in nb_mvdr nb_mvdr_m4; do
  for a in "--steps 20 --warmup 5" "--steps 2 --warmup 1 --hops-per-step 625"; do
    echo -n "$c $a  "; timeout 300 python bench.py --config $c $a --no-cpu-baseline --no-extras 2>$O/err_$c.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['bytes_per_launch'])"
  done
done
PROFILE_SQ=1 bash scripts/profile_bench.sh r05d_nb_mvdr --config nb_mvdr > $O/prof_nb_mvdr.txt 2>&1; head -8 $O/prof_nb_mvdr.txt; grep -A12 "HBM traffic" $O/prof_nb_mvdr.txt | head -8; grep -B1 -A12 "pmc_sq: void ds::ds_binop_kernel<8, 6>" $O/prof_nb_mvdr.txt | head -60
