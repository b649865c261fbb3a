#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_job06
timeout 600 python scratch/dbg_nan3.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_job06/dbg3b.log
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_wpe_wide.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_job06/pytest.log
