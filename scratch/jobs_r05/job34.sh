#!/bin/bash
# round 5: the Nyquist bin of the plain MVDR kernels as a real-valued program (adaptive_bin_nyq): parity tests, cfg2 A/B in both regimes (nyqgen = the general program, -DDS_NYQ_REAL=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ny; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity.jsonl timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
for r in "T1 --steps 20 --warmup 5" "T625 --steps 2 --warmup 1 --hops-per-step 625" "T16 --steps 4 --warmup 2 --hops-per-step 16"; do set -- $r; n=$1; shift; for i in 1 2 3; do for v in nyqgen nyqreal; do
  echo -n "cfg2_$n $v  "; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config cfg2 "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/nyq_real_ab.txt
