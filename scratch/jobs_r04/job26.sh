#!/bin/bash
# round 4, job 26: block_rw, the register-burst pattern with arithmetic between the loads and the stores, at 4 / 8 / 16 waves per SIMD-set
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job26; mkdir -p $O
timeout 600 scratch/micro/block_rw 132096 spin 2>&1 | tee $O/block_rw_spin.txt
