#!/bin/bash
O=gpurun_out/r06l; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default_k20.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; wc -c $O/bench_default_k20.json; cp bench_detail.json $O/bench_detail.json
python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
