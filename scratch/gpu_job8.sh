#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "== current run $i"; timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "utterance_groups_equal" 2>&1 | tail -2
done
for i in 1 2 3; do
  echo "== base variant run $i"; DSENH_LIB=$GRAFT_REPO_ROOT/scratch/variants/libdsenh_base.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "utterance_groups_equal" 2>&1 | tail -2
done
