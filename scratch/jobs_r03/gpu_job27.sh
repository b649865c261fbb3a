#!/bin/bash
# cfg5: is the step length bistable within a run or between runs?  four traced runs of 300 steps each
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 DS_BENCH_SYNTH=white
for i in 1 2 3 4; do
  O=$GRAFT_REPO_ROOT/gpurun_out/iv$i; rm -rf $O; mkdir -p $O
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 300 --warmup 10 --min-region-ms 20 --no-cpu-baseline --no-extras > $O/trace.log 2>&1)
  echo "== run $i: $(grep -o '"ms_per_step": [0-9.]*' $O/trace.log | head -1)"
  python3 scratch/cfg5_intervals.py $O
  rm -rf $O/trace
done
