#!/bin/bash
# round 4, check of the committed tree: the GPU suite with its parity log (default library), the shelved experiments' tests on their own
# build, smoke(), the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/libdsenh_shelved.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ops.py -q -m gpu -k "quad_kernel or pipelined_kernel or fused_tail_and_pipelined" > $O/gpu_tests_shelved.txt 2>&1; tail -2 $O/gpu_tests_shelved.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/bench_detail.json
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default_k20.json 2> $O/bench_default_k20.err
wc -c $O/bench_default_k20.json; tail -4 $O/bench_default_k20.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04m/bench_default_k20.json').read().strip().splitlines()[-1])
r=d['roofline']; print('cfg2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'measured', r.get('frac_measured'), r.get('traffic_profile'))
print('hbm', d['roofline_hbm']['value'], d['roofline_hbm']['frac'], d['roofline_hbm'].get('frac_measured'))
for k,v in d['other_configs'].items(): print(k, v)
print(d.get('latency_us'), d.get('cpu_baseline'))
PY
