#!/bin/bash
# round 5: two-row WPE kernel with the prediction filters as rows per lane (no partial-product array / err hand-off in LDS; one phase less per frame): WPE tests, cfg4 A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05wf; mkdir -p $O
DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_wrows_final.so timeout 1200 python -m pytest tests -m gpu -q -k "wpe or cfg4 or chain or dereverb" 2>&1 | tail -3
for T in 8 16 64 312; do for i in 1 2 3; do for v in head6 wrows_final; do
    echo -n "T$T $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config cfg4 --steps 4 --warmup 2 --hops-per-step $T --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done 2>&1 | tee $O/cfg4_wpe2_wrows_ab.txt
