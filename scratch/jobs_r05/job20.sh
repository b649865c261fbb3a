#!/bin/bash
# round 5: two-row WPE kernel with its LDS rows padded against bank conflicts across the bins of a wavefront: cfg4 A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x; mkdir -p $O
for T in 8 16 64 312; do for i in 1 2 3; do for v in mvdrc wpe2pad; do
    echo -n "T$T $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config cfg4 --steps 4 --warmup 2 --hops-per-step $T --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done > $O/cfg4_wpe2_pad_ab.txt 2>&1
cat $O/cfg4_wpe2_pad_ab.txt
