#!/bin/bash
# cfg4 (WPE -> McMcra -> MVDR, 8 mics, 1024-point frames, B = 1024), 10 s per call: the chain WITHOUT one stage's launch at a time
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06abl; mkdir -p $O
export DS_BENCH_SYNTH=white DSENH_LIB=$R/scratch/libdsenh_abl.so
A="--config cfg4 --hops-per-step 312 --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for s in none none stft wpe mcmcra mvdr istft "stft,mcmcra,mvdr,istft" "mcmcra,mvdr"; do
  DS_ABL_SKIP=$s python3 $R/bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('skip=%-24s %6.2f M frames/s  %7.2f ms per 312-block call' % ('$s', d['value']/1e6, d['ms_per_step']))" | tee -a $O/cfg4_stage_ablation.txt
done
