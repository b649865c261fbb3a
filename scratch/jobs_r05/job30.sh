#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k2; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity.jsonl timeout 900 python -m pytest tests/test_gpu_wpe_wide.py -m gpu -q -k "known_reverberation" 2>&1 | tail -12; cat $O/parity.jsonl
