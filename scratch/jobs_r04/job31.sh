#!/bin/bash
# round 4, job 31: the frame kernels' state traffic ordinary (libdsenh_plain.so, -DDS_PLAIN_STATE) against non-temporal (libdsenh.so) by batch size:
# does the non-temporal policy keep a small state out of the Infinity Cache?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job31; mkdir -p $O
for rep in 1 2; do
for lib in libdsenh.so libdsenh_plain.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for b in 256 1024 2048 4096 16384; do
    timeout 600 python bench.py --config cfg2 --batch $b --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg2 B=$b', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config cfg3 --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg3', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
done
