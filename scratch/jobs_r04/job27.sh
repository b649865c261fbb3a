#!/bin/bash
# round 4, job 27: SubbandGSC(postfilter=True).omlsa_multi against the reference's G22 fixtures
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job27; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1800 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "subband_gsc" 2>&1 | tail -30 | tee -a $O/pytest.log
