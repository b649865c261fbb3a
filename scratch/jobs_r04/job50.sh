#!/bin/bash
# round 4, job 50: the traffic guards with their final tolerances
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job50; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -q -m gpu -k "traffic or budget or bench" 2>&1 | tail -4 | tee -a $O/pytest.log
