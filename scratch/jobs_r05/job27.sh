#!/bin/bash
# round 5: the cfg5 chain's piece length (blocks per pipeline piece; 62 since round 3) again, now that the McSpp operator is a fifth shorter
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p2; mkdir -p $O
for i in 1 2; do for v in piece32 piece44 piece62 piece90 piece125; do
  echo -n "cfg5_T625 $v  "
  DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_$v.so timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 --hops-per-step 625 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done > $O/cfg5_piece_sweep.txt 2>&1
cat $O/cfg5_piece_sweep.txt
