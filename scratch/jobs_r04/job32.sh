#!/bin/bash
# round 4, job 32: (a) frame-kernel plane rows of 264 lanes (libdsenh_kp8.so) against 260 (libdsenh.so); (b) the wide WPE block on 128-byte lines
# (libdsenh.so) against the packed one (libdsenh_shelved.so = the tree two commits back)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job32; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/pytest.log
for rep in 1 2; do
for lib in libdsenh.so libdsenh_kp8.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for b in 1024 4096 16384; do
    timeout 600 python bench.py --config cfg2 --batch $b --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg2 B=$b', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config cfg3 --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg3', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
for lib in libdsenh_shelved.so libdsenh.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in wpe_nb cfg4_n10; do
    timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
done
done
