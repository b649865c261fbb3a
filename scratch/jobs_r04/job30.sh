#!/bin/bash
# round 4, job 30: plane rows of 260 lanes (32.5 lines per 16-byte plane row) against 264 (33 lines): scratch/micro/planes_bw.hip;
# and block_rw with the wide WPE block on a line multiple (29184 B against 29120 B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job30; mkdir -p $O
( echo "== KP 260"; timeout 300 scratch/micro/planes_bw; echo "== KP 264"; timeout 300 scratch/micro/planes_bw_264 ) 2>&1 | grep -E "==|^B =|16-byte" | tee $O/planes_kp.txt
( echo "== block 29120"; timeout 300 scratch/micro/block_rw 132096 | grep -E "chunks, in place  |nt loads\+stores|whole block, in place" ; echo "== block 29184"; timeout 300 scratch/micro/block_rw 132096 blk 29184 | grep -E "chunks, in place  |nt loads\+stores|whole block, in place" ) 2>&1 | tee $O/block_align.txt
