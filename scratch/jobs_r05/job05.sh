#!/bin/bash
# round 5: the DC notch in double (G22 at ten times the level: 1.0e-4 -> ?), the whole GPU suite with its parity log, cfg5 / TDGSC / FDGSC rates
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
grep -h "G22\|G12\|G15\|G16\|G17_sub\|online_mvdr" $O/parity_measured.jsonl | cut -c1-400
for c in cfg5 tdgsc fdgsc; do
  echo -n "$c T1  "; timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
echo -n "cfg5 T625  "; timeout 300 python bench.py --config cfg5 --steps 2 --warmup 1 --hops-per-step 625 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
