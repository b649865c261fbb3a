#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O
python bench.py --host-api 2>&1 | tail -1 > $O/host_api.json; python -c "
import sys,json
d=json.load(open('$O/host_api.json'))
print('pinned h2d %.1f d2h %.1f GB/s' % (d['pinned_h2d_gbs'], d['pinned_d2h_gbs']))
for T in ('T4','T625'):
    for k in ('mirror','mirror_f32','c_abi','pcm16'):
        e=d[T][k]; print(T,k,'%.2f M frames/s  %.3f ms  %.1f GB/s  frac %.2f' % (e['frames_s']/1e6, e['ms_per_call'], e['gbs'], e['frac_of_pinned']))
"
python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
