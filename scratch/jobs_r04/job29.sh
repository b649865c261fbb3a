#!/bin/bash
# round 4, job 29: RLS-WPE blocks of 16 taps-by-channels on 128-byte lines + non-temporal block traffic (libdsenh.so) against the packed
# block with ordinary accesses (libdsenh_shelved.so = the tree before): WPE / chain tests, then cfg4 three times each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job29; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wpe_wide.py tests/test_gpu_parity.py -x -q -m gpu -k "wpe or Wpe or WPE or chain or cfg4 or dereverb" 2>&1 | tail -12 | tee -a $O/pytest.log
for rep in 1 2 3; do
for lib in libdsenh_shelved.so libdsenh.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  timeout 600 python bench.py --config cfg4 --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg4 T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  timeout 600 python bench.py --config cfg4 --steps 3 --warmup 1 --hops-per-step 312 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg4 T=312', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
done
