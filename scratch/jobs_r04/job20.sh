#!/bin/bash
# round 4, job 20: the operators' state buffer accesses non-temporal (libdsenh.so) against ordinary (libdsenh_ops0.so), three rounds each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job20; mkdir -p $O
for rep in 1 2 3; do
for lib in libdsenh_ops0.so libdsenh.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in cfg4 cfg5; do
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config cfg4 --steps 3 --warmup 1 --hops-per-step 312 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg4 T=312', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
  timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --hops-per-step 625 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg5 T=625', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
done
