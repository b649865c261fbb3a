#!/bin/bash
# round 4, job 43: the operators' state LOADS streamed and the stores ordinary (libdsenh_opsld.so) against both streamed (libdsenh.so): speed
# (five interleaved rounds) and the HBM write bytes of a cfg5 step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job43; mkdir -p $O
for rep in 1 2 3 4 5; do
for lib in libdsenh.so libdsenh_opsld.so; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/$lib
  for cfg in cfg5 cfg4; do
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
done
done
export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/libdsenh_opsld.so
bash scripts/profile_bench.sh r04_j43_opsld_cfg5 --config cfg5 --steps 20 > /dev/null 2>&1
bash scripts/profile_bench.sh r04_j43_opsld_cfg4 --config cfg4 --steps 20 > /dev/null 2>&1
for c in cfg5 cfg4; do cp gpurun_out/prof_r04_j43_opsld_$c/traffic.json $O/${c}_opsld_traffic.json; rm -rf gpurun_out/prof_r04_j43_opsld_$c/trace gpurun_out/prof_r04_j43_opsld_$c/pmc_*/; done
python - <<'PY'
import json
for c in ('cfg5','cfg4'):
    t=json.load(open('gpurun_out/r04_job43/%s_opsld_traffic.json'%c)); print(c, 'opsld step MB', round(t['hbm_bytes_per_step']/1e6,1))
    for k,v in t['kernels'].items(): print('   %-60s fetch %8.1f write %8.1f' % (k[:60], v['fetch_bytes']/1e6, v['write_bytes']/1e6))
PY
