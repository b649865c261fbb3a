#!/bin/bash
# round 4, job 44: the operators' 16-byte groups as ONE buffer_store_dwordx4 (inline asm; libdsenh.so) against dword + dwordx3 (libdsenh_prest4.so):
# tests (operators, chains, two-stream groups), speed (five interleaved rounds), write bytes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job44; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -4 | tee -a $O/pytest.log
for rep in 1 2 3 4 5; do
for lib in libdsenh.so libdsenh_prest4.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in cfg5 cfg4; do
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
done
done
unset DSENH_LIB
bash scripts/profile_bench.sh r04_j44_cfg5 --config cfg5 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh r04_j44_cfg4 --config cfg4 --steps 20 > /dev/null 2>&1
for c in cfg5 cfg4; do cp gpurun_out/prof_r04_j44_$c/traffic.json $O/${c}_traffic.json; cp gpurun_out/prof_r04_j44_$c/summary.txt $O/${c}_summary.txt; cp gpurun_out/prof_r04_j44_$c/kernel_stats.csv $O/${c}_kernel_stats.csv; f=$(find gpurun_out/prof_r04_j44_$c/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${c}_rocprofv3_stats.csv; rm -rf gpurun_out/prof_r04_j44_$c/trace gpurun_out/prof_r04_j44_$c/pmc_*/; done
python - <<'PY'
import json
for c in ('cfg5','cfg4'):
    t=json.load(open('gpurun_out/r04_job44/%s_traffic.json'%c)); print(c, 'step MB', round(t['hbm_bytes_per_step']/1e6,1))
    for k,v in t['kernels'].items(): print('   %-60s fetch %8.1f write %8.1f' % (k[:60], v['fetch_bytes']/1e6, v['write_bytes']/1e6))
PY
