#!/bin/bash
# cfg5, 10 s per call: the chain's four streams on disjoint sets of CUs (hipExtStreamCreateWithCUMask; bit ranges of the 256-bit mask) against
# all of them on every CU.  Stream bits: 1 McSpp's, 2 the tail's, 4 the front end's, 8 the blocking filters'.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
A="--config cfg5 --hops-per-step 625 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
run() { env "$@" python3 $R/bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-72s %6.2f M frames/s  %7.2f ms' % ('$*', d['value']/1e6, d['ms_per_step']))" | tee -a $O/cfg5_cumask_ab.txt; }
run X=0
run DS_ABL_CU_4=0-104 DS_ABL_CU_2=104-168 DS_ABL_CU_1=168-216 DS_ABL_CU_8=216-256
run DS_ABL_CU_4=0-112 DS_ABL_CU_2=112-184 DS_ABL_CU_1=184-256 DS_ABL_CU_8=184-256
run DS_ABL_CU_4=0-128 DS_ABL_CU_2=128-256 DS_ABL_CU_1=128-256 DS_ABL_CU_8=0-128
run DS_ABL_CU_4=0-160 DS_ABL_CU_2=96-256
run DS_ABL_CU_2=0-128
run X=0
