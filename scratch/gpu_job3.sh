#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pipelined or full_batch_properties or golden" > gpurun_out/r03b/gpu_tests_pipe.txt 2>&1
tail -5 gpurun_out/r03b/gpu_tests_pipe.txt
timeout 900 python scratch/perf_pipe_ab.py > gpurun_out/r03b/pipe_ab.txt 2>&1
cat gpurun_out/r03b/pipe_ab.txt
