#!/bin/bash
# round 4, job 22: GSC reference powers (DS_PARAM_REF_POWERS) and the omlsa_multi mirror; cfg3 before / after (libdsenh_shelved.so is the previous tree)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job22; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gsc or ref_powers" 2>&1 | tail -40 | tee -a $O/pytest.log
for rep in 1 2 3; do
for lib in libdsenh_shelved.so libdsenh.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  timeout 600 python bench.py --config cfg3 --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg3 T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  timeout 600 python bench.py --config cfg3 --steps 3 --warmup 1 --hops-per-step 625 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib cfg3 T=625', d['value'], d['ms_per_step'])" | tee -a $O/bench.log
done
done
