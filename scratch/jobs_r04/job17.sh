#!/bin/bash
# round 4, job 17: which non-temporal accesses pay where.  plain = no nt in the WPE / operator kernels; ntst = nt stores only (WPE);
# ntld = nt loads only (WPE, LDS-DMA included); wpent = both (WPE); nt = both + the operators' state buffer accesses
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_job17; mkdir -p $O
for lib in plain ntst ntld wpent nt; do
  export DSENH_LIB=$GRAFT_REPO_ROOT/distantspeech_amd/libdsenh_$lib.so
  for cfg in wpe_nb cfg4_n10 cfg4 cfg5; do
    timeout 600 python bench.py --config $cfg --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $cfg T=1', d['value'], d['ms_per_step'], d['roofline']['frac'])" | tee -a $O/bench.log
  done
done
