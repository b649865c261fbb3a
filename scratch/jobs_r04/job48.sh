#!/bin/bash
# round 4, job 48: soak of the last build: the whole GPU suite three times, the two-stream group tests twenty times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job48; mkdir -p $O
for i in 1 2 3; do timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -1 | tee -a $O/soak.log; done
for i in $(seq 1 20); do timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "utterance_groups or sequences_graphs" 2>&1 | tail -1 | tee -a $O/soak_groups.log; done
sort $O/soak_groups.log | uniq -c
