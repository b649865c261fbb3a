#!/bin/bash
# round 4, job 1: wide-tap WPE — parity tests, then a first bench of the notebook operating point
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_job01
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r04_job01/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04_job01/pytest_wide.log
cat gpurun_out/r04_job01/pytest_wide.log
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wpe or chain_stage" 2>&1 | tail -8 > gpurun_out/r04_job01/pytest_wpe_old.log
cat gpurun_out/r04_job01/pytest_wpe_old.log
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/gpurun_out/r04_job01/detail_wpe_nb.json
timeout 600 python bench.py --config wpe_nb --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/r04_job01/bench_wpe_nb.log
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/gpurun_out/r04_job01/detail_wpe_nb_T250.json
timeout 600 python bench.py --config wpe_nb --steps 2 --warmup 1 --hops-per-step 250 --no-extras --no-cpu-baseline 2>&1 | tail -3 | tee gpurun_out/r04_job01/bench_wpe_nb_T250.log
