#!/bin/bash
# round 5: Hermitian rank-1 update and the Cholesky factor / products without conjugates built in register pairs (McSpp steady operator 1430 -> 1131 vector instructions): GPU suite, A/B on cfg5 / nb_mvdr / cfg2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
ab() {
  for i in 1 2 3; do for v in wpe2_t2 nomov; do
    echo -n "$1 $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config $3 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
}
( ab cfg5_T1 "--steps 20 --warmup 5" cfg5; ab cfg5_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg5; ab nb_mvdr_T1 "--steps 20 --warmup 5" nb_mvdr; ab cfg2_T1 "--steps 20 --warmup 5" cfg2; ab cfg2_T625 "--steps 2 --warmup 1 --hops-per-step 625" cfg2 ) > $O/nomov_ab.txt 2>&1
cat $O/nomov_ab.txt
