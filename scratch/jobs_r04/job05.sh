#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job05; mkdir -p $O
DS_FORCE_DEVICE=0 DS_DIST_BACKEND=gloo DS_BENCH_DETAIL=$O/d.json timeout 600 python bench.py --gpus 2 --config cfg4 --total-batch 96 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --min-region-ms 20 > $O/out.log 2> $O/err.log
echo rc=$?; grep -v "^\[rank0\]" $O/err.log | head -60; tail -2 $O/out.log
