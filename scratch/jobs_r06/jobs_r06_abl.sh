#!/bin/bash
# cfg5, 10 s per call: the pipelined chain WITHOUT one stage's launch at a time (library built with -DDS_ABLATE_CHAIN, DS_ABL_SKIP=<stage>):
# what each stage costs the chain next to the others.  Results are garbage by construction (a stage's output is stale); timing only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
A="--config cfg5 --hops-per-step 625 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for s in none none notch fir cdr mcspp rows fan tail "notch,fir,cdr" "fan,rows" "mcspp,fan,rows"; do
  DS_ABL_SKIP=$s python3 $R/bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('skip=%-16s %6.2f M frames/s  %7.2f ms per 625-block call' % ('$s', d['value']/1e6, d['ms_per_step']))" | tee -a $O/cfg5_stage_ablation.txt
done
