#!/bin/bash
# the DC notch's tile shape (32 rows x 256 samples against 64 x 128) by call length: cfg5 chain, tdgsc (its other user)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
run() { c=$1; t=$2; shift 2; env "$@" python3 $R/bench.py --config $c --hops-per-step $t --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-6s T=%-4s %-20s %7.3f M frames/s  %9.4f ms per step' % ('$c', '$t', '$*', d['value']/1e6, d['ms_per_step']))" | tee -a $O/notch_shape_ab.txt; }
for rep in 1 2; do
for t in 1 4 62; do
run cfg5 $t X=0
run cfg5 $t DS_ABL_NOTCH64=1
done
run tdgsc 1 X=0
run tdgsc 1 DS_ABL_NOTCH64=1
done
