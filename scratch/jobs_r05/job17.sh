#!/bin/bash
# round 5: where the two-row WPE kernel starts to pay: cfg4 at T = 2 .. 64 hops per call, one-row kernel (r05_head) against two-row from T = 2 on
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
for T in 2 4 8 16 64 312; do for i in 1 2; do for v in r05_head wpe2_t2; do
    echo -n "T$T $v  "
    DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 300 python bench.py --config cfg4 --steps 4 --warmup 2 --hops-per-step $T --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done; done > $O/cfg4_wpe2_tsweep.txt 2>&1
cat $O/cfg4_wpe2_tsweep.txt
