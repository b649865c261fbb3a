#!/bin/bash
# round 4, job 4: full GPU suite on the default library, the shelved experiments' tests on their own build, the driver's bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job04; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_gpu.log
DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/libdsenh_shelved.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad_kernel or pipelined_kernel" 2>&1 | tail -4 | tee $O/pytest_shelved.log
export DS_BENCH_DETAIL=$GRAFT_REPO_ROOT/$O/bench_detail.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.log 2> $O/bench_default.err
tail -c 4500 $O/bench_default.log; tail -5 $O/bench_default.err; wc -c $O/bench_default.log
