#!/bin/bash
# round 5: the narrow WPE kernel with two rows of P per lane (ds_wpe2.hpp; bit-identical to the one-row program), with / without hoisted lane geometry: WPE tests, cfg4 A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_wpe2_h0.so timeout 1500 python -m pytest tests -m gpu -q -k "wpe or cfg4 or chain or dereverb" > $O/gpu_tests_wpe.txt 2>&1; tail -6 $O/gpu_tests_wpe.txt
ab() {
  for i in 1 2 3; do for v in r05_head wpe2_h0 wpe2_h2; do
    echo -n "$1 $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config cfg4 $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done; done
}
( ab T1 "--steps 20 --warmup 5"; ab T312 "--steps 2 --warmup 1 --hops-per-step 312" ) > $O/cfg4_wpe2_ab.txt 2>&1
cat $O/cfg4_wpe2_ab.txt
