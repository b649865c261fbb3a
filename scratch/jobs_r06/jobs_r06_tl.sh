#!/bin/bash
# timeline of the cfg5 10 s call (62-block pieces): which launches wait for which
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06tl; mkdir -p $O
export DS_BENCH_SYNTH=white
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --config cfg5 --hops-per-step 625 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/trace.log 2>&1
python3 $R/scripts/timeline_rows.py /tmp/tl 90 > $O/cfg5_T625_timeline.txt 2>&1
tail -3 $O/trace.log
head -40 $O/cfg5_T625_timeline.txt
