#!/bin/bash
# round 4, job 15: the wide-WPE access pattern without arithmetic (scratch/micro/block_rw.hip)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job15; mkdir -p $O
timeout 600 scratch/micro/block_rw 2>&1 | tee $O/block_rw.txt
