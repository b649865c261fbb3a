#!/bin/bash
# round 4, job 49: what the Nyquist bin's per-bin program costs a hop: the frame kernels WITHOUT it (libdsenh_nonyq.so, -DDS_ABLATE_NYQUIST: wrong
# results, timing only) against the product (libdsenh.so), 10 s per call and one hop per call, three interleaved rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job49; mkdir -p $O
for rep in 1 2 3; do
for lib in libdsenh.so libdsenh_nonyq.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in cfg2 cfg3; do
    timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --hops-per-step 625 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=625', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config cfg2 --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg2 T=1', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
done
