#!/bin/bash
# the committed build once more: GPU suite (parity log), smoke, the driver's bench command
O=gpurun_out/${1:-r06m}; mkdir -p $O
rm -f gpurun_out/parity_measured.jsonl
python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench_default_k20.json 2> $O/bench_default.err; tail -2 $O/bench_default.err; wc -c $O/bench_default_k20.json; cp bench_detail.json $O/bench_detail.json
cp gpurun_out/parity_measured.jsonl $O/parity_measured.jsonl
