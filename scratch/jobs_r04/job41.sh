#!/bin/bash
# round 4, job 41: the frame kernels' plane traffic: streamed both ways (libdsenh.so), streamed stores only (ntst), streamed loads only (ntld), by batch size
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job41; mkdir -p $O
for rep in 1 2 3; do
for lib in libdsenh.so libdsenh_ntst.so libdsenh_ntld.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for b in 1024 4096 16384; do
    timeout 600 python bench.py --config cfg2 --batch $b --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg2 B=$b', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config cfg3 --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg3', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
done
