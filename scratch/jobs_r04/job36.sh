#!/bin/bash
# round 4, job 36: the operators' state accesses non-temporal (libdsenh_opsnt.so) against ordinary (libdsenh.so) now that the plane rows are
# on 128-byte lines (job 20 had found no gain on the 260-lane rows); three rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job36; mkdir -p $O
for rep in 1 2 3; do
for lib in libdsenh.so libdsenh_opsnt.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in cfg4 cfg5; do
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
done
done
