#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t2; mkdir -p $O
DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity.jsonl timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "notebook_online_mvdr" 2>&1 | tail -5; cat $O/parity.jsonl | tail -3
