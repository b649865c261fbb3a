#!/bin/bash
# cfg5, 10 s per call: the piece length again (62 blocks was the best of round 2's kernels), with this round's kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
A="--config cfg5 --hops-per-step 625 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for rep in 1 2; do for q in 62 24 32 48 78 104 125 157; do
  DS_ABL_PIECE=$q python3 $R/bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('piece=%-4s %6.2f M frames/s  %7.2f ms per 625-block call' % ('$q', d['value']/1e6, d['ms_per_step']))" | tee -a $O/cfg5_piece_sweep.txt
done; done
