#!/bin/bash
# round 5: the whole GPU suite (two-row WPE kernel, late-staged 8-microphone GSC kernel)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 900 python -m pytest tests/test_gpu_wpe_wide.py -m gpu -q -k "double" > $O/gpu_tests_wpe.txt 2>&1; tail -6 $O/gpu_tests_wpe.txt
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
grep -h "fp64" $O/parity_measured.jsonl | cut -c1-300
