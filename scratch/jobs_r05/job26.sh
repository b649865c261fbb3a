#!/bin/bash
# round 5: the shelved build (make SHELVED=1: hop-pipelined frame kernels, quad-lane 8-microphone kernels, fused front end, fused fan) against this tree: its 9 GPU tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05sh; mkdir -p $O
DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_shelved_r05.so timeout 1500 python -m pytest tests -m gpu -q -k "quad or pipelin or shelved or front_fused or fan_fused or chain_variants" > $O/gpu_tests_shelved.txt 2>&1; tail -5 $O/gpu_tests_shelved.txt
