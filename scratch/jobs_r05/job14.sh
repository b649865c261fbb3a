#!/bin/bash
# round 5: 8 microphones at 1024 points with the input staged global -> LDS during the inverse transform (GSC kernel: scratch 124 B -> 0): GPU suite, A/B against HEAD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -6 $O/gpu_tests.txt
( for a in 2 1 0; do for i in 1 2; do echo "algo $a"; DS_SHAPE_ALGO=$a DS_SHAPE_B=512 python scratch/perf_shape_one.py 8 1024 r05_head late; done; done ) > $O/m8_1024_late_ab.txt 2>&1
cat $O/m8_1024_late_ab.txt
