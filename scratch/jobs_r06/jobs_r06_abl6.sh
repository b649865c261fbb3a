#!/bin/bash
# stage ablation with real (stale) data: every stage runs during the warm-up call (DS_ABL_AFTER = the pieces / group calls of one bench step),
# the named stage's launch is skipped from then on.  cfg4: 1 chain call per step; cfg5: 11 pieces per 625-block step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
run() { c=$1; t=$2; a=$3; s=$4; DS_ABL_AFTER=$a DS_ABL_SKIP=$s python3 $R/bench.py --config $c --hops-per-step $t --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-5s skip=%-24s %6.2f M frames/s  %7.2f ms per call' % ('$c', '$s', d['value']/1e6, d['ms_per_step']))" | tee -a $O/stage_ablation_stale.txt; }
for s in none stft wpe mcmcra mvdr istft "stft,mcmcra,mvdr,istft"; do run cfg4 312 1 $s; done
for s in none notch fir cdr mcspp rows fan tail "notch,fir,cdr" "mcspp,fan,rows"; do run cfg5 625 11 $s; done
