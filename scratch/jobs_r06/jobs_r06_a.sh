#!/bin/bash
# round 6, first GPU job: the direct eigen-solve of the notebook operator — parity tests of everything that uses it, then the two bench workloads
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "mcspp or notebook or steering or gev" > $O/tests_mcspp.txt 2>&1; tail -3 $O/tests_mcspp.txt
for c in nb_mvdr nb_mvdr_m4; do
  python bench.py --config $c --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; tail -c 600 $O/bench_$c.json
  python bench.py --config $c --hops-per-step 625 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_${c}_T625.json 2>> $O/bench_$c.err; tail -c 300 $O/bench_${c}_T625.json
done
