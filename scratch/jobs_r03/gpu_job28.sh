#!/bin/bash
# A/B: prev = the Nyquist bin as a second pass of the per-bin phase for 256- / 1024-point frames; new = its pass next to the first inverse stage for every frame size
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== prev"; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_prev.so timeout 900 python scratch/perf_shapes.py 2>&1 | tail -24
echo "== new"; timeout 900 python scratch/perf_shapes.py 2>&1 | tail -24
