#!/bin/bash
# round 4, job 23: the GSC reference-power tests again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job23; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gsc or ref_powers" 2>&1 | tail -30 | tee -a $O/pytest.log
