#!/bin/bash
# round 4, job 28: block_rw, the narrow WPE kernel's access shape: 2244-byte blocks against line-aligned 2304-byte blocks, plain against nt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job28; mkdir -p $O
timeout 600 scratch/micro/block_rw 132096 narrow 2>&1 | tee $O/block_rw_narrow.txt
