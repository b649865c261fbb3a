#!/bin/bash
# round 4, job 14: lds_load_wait() as the s_waitcnt builtin (the compiler sees the LDS-DMA land), in both wide-WPE forms:
# libdsenh.so = one workgroup per bin, two lanes per row; libdsenh_onewave.so = one wavefront per bin (committed form) + the first chunk's
# copies issued in front of the other loads.  Before (inline-asm wait): one wavefront 680 k / 149 k / 3.53 M, half rows 633 k / 146 k / 2.39 M
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job14; mkdir -p $O
for lib in libdsenh_onewave.so libdsenh.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  timeout 900 python -m pytest tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/pytest_wide.log
  for cfg in wpe_nb cfg4_n10; do
    timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
  timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib wpe_nb T=250', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
