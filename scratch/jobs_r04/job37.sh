#!/bin/bash
# round 4, job 37: job 36 again, five rounds, with the 10 s-per-call workloads (do streamed state lines hurt when the operators come back to their state?)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job37; mkdir -p $O
for rep in 1 2 3 4 5; do
for lib in libdsenh_opsnt.so libdsenh.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in cfg5 cfg4; do
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
done
done
for lib in libdsenh_opsnt.so libdsenh.so libdsenh_opsnt.so libdsenh.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --hops-per-step 625 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg5 T=625', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
  timeout 600 python bench.py --config cfg4 --steps 3 --warmup 1 --hops-per-step 312 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg4 T=312', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
timeout 1800 env DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/libdsenh_opsnt.so python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/pytest_opsnt.log
