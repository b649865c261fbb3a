#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O
./scratch/micro/host_path > $O/host_path.txt 2>&1; cat $O/host_path.txt
python -m pytest tests/test_gpu_bench.py tests/test_gpu_parity.py -q -m gpu -k "bench or rank or verify or rccl or beampattern or estpos" > $O/gpu_bench_tests.txt 2>&1; tail -5 $O/gpu_bench_tests.txt
