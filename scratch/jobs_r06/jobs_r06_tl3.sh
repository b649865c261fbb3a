#!/bin/bash
# cfg4, 10 s per call: kernel rows of the grouped chain
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06tl; mkdir -p $O
export DS_BENCH_SYNTH=white GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl4 -- python3 $R/bench.py --config cfg4 --hops-per-step 312 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/trace_cfg4.log 2>&1
python3 $R/scripts/timeline_rows.py /tmp/tl4 40 > $O/cfg4_T312_rows.txt 2>&1
cat $O/cfg4_T312_rows.txt
