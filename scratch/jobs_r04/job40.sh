#!/bin/bash
# round 4, job 40: the wide WPE kernel with W, the taps and var through the tile as well (three chunks; libdsenh.so) against two chunks + direct
# strided accesses for them (libdsenh_pretail.so), five interleaved rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job40; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/pytest.log
for rep in 1 2 3 4 5; do
for lib in libdsenh.so libdsenh_pretail.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in wpe_nb cfg4_n10; do
    timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
done
done
for lib in libdsenh.so libdsenh_pretail.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib wpe_nb T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
