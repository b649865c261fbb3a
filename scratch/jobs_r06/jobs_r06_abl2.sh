#!/bin/bash
# cfg5, 10 s per call: kernel-shape experiments of the ablation build (timing only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
A="--config cfg5 --hops-per-step 625 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
run() { env "$@" python3 $R/bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-40s %6.2f M frames/s  %7.2f ms per 625-block call' % ('$*', d['value']/1e6, d['ms_per_step']))" | tee -a $O/cfg5_shape_ab.txt; }
for rep in 1 2; do
run X=0
run DS_ABL_FIR_OPL4=1
run DS_ABL_NOTCH64=1
run DS_ABL_FIR_OPL4=1 DS_ABL_NOTCH64=1
done
