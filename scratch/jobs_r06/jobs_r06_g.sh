#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
for g in 1 0 2 4 7; do
  if [ $g = 0 ]; then unset DS_IO_GROUPS; else export DS_IO_GROUPS=$g; fi
  echo "DS_IO_GROUPS=$g"; python bench.py --host-api 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('pinned h2d %.1f d2h %.1f GB/s' % (d['pinned_h2d_gbs'], d['pinned_d2h_gbs']))
for T in ('T4','T625'):
    for k in ('mirror','c_abi','pcm16'):
        e=d[T][k]; print(T,k,'%.2f M frames/s  %.3f ms  %.1f GB/s  frac %.2f' % (e['frames_s']/1e6, e['ms_per_call'], e['gbs'], e['frac_of_pinned']))
"
done 2>&1 | tee $O/host_api_ab.txt
unset DS_IO_GROUPS
python -m pytest tests -q -m gpu -x > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
