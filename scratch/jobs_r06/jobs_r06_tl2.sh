#!/bin/bash
# cfg5 10 s call: does the number of hardware queues bound the stage pipeline?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06tl; mkdir -p $O
export DS_BENCH_SYNTH=white
A="--config cfg5 --hops-per-step 625 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for q in 8; do
  GPU_MAX_HW_QUEUES=$q python3 $R/bench.py $A 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('hwq=$q', d['value'], d['ms_per_step'])" | tee -a $O/hwq.txt
done
export GPU_MAX_HW_QUEUES=8
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl8 -- python3 $R/bench.py --config cfg5 --hops-per-step 625 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/trace8.log 2>&1
python3 $R/scripts/timeline_rows.py /tmp/tl8 60 > $O/cfg5_T625_timeline_hwq8.txt 2>&1
head -30 $O/cfg5_T625_timeline_hwq8.txt
