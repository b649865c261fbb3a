#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
python -m pytest tests -q -m gpu -k "tdgsc or fdgsc or gsc_chain or realtime or pcm16 or host or wav_files or fixed" > $O/gpu_tests_subset.txt 2>&1; tail -3 $O/gpu_tests_subset.txt
python bench.py --host-api 2>&1 | tail -1 > $O/host_api.json; python -c "
import sys,json
d=json.load(open('$O/host_api.json'))
print('pinned h2d %.1f d2h %.1f GB/s' % (d['pinned_h2d_gbs'], d['pinned_d2h_gbs']))
for T in ('T4','T625'):
    for k in ('mirror','mirror_f32','c_abi','pcm16'):
        e=d[T][k]; print(T,k,'%.2f M frames/s  %.3f ms  %.1f GB/s  frac %.2f' % (e['frames_s']/1e6, e['ms_per_call'], e['gbs'], e['frac_of_pinned']))
" | tee $O/host_api.txt
for c in tdgsc fdgsc; do
  echo -n "$c "; python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
  PROFILE_HBM=0 bash scripts/profile_bench.sh r06i_$c --config $c > /dev/null 2>&1; cp gpurun_out/prof_r06i_$c/kernel_stats.csv $O/${c}_kernel_stats.csv; head -12 $O/${c}_kernel_stats.csv; rm -rf gpurun_out/prof_r06i_$c/trace
done 2>&1 | tee $O/gsc_chains.txt
DS_PARITY_LOG=$PWD/$O/r06_wpe_sample.jsonl python scratch/wpe_sample.py 32 2>&1 | tail -8 | cut -c1-600
