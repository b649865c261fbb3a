#!/bin/bash
# round 4, job 38: the cfg5 tail kernel's canceller planes streamed (libdsenh.so) against ordinary (libdsenh_aicplain.so = the commit before), five rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job38; mkdir -p $O
for rep in 1 2 3 4 5; do
for lib in libdsenh.so libdsenh_aicplain.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  timeout 600 python bench.py --config cfg5 --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg5 T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
done
done
for lib in libdsenh.so libdsenh_aicplain.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --hops-per-step 625 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg5 T=625', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -x -q -m gpu -k "subband_gsc or chain" 2>&1 | tail -3 | tee -a $O/pytest.log
