#!/bin/bash
# round 5: the RLS-WPE recursion in double (DS_PARAM_WPE_FP64, ds_wpe64.hpp): its tests, the other WPE tests, and its rate beside the fp32 kernel's
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
export DS_PARITY_LOG=$GRAFT_REPO_ROOT/$O/parity_measured.jsonl
timeout 1500 python -m pytest tests/test_gpu_wpe_wide.py tests/test_gpu_ops.py -m gpu -q -x -k "wpe" > $O/gpu_tests_wpe.txt 2>&1; tail -6 $O/gpu_tests_wpe.txt
grep -h "fp64\|wpe_wide_32s" $O/parity_measured.jsonl | cut -c1-300
python - <<'PY' 2>&1 | tail -6
import time, numpy as np
import distantspeech_amd as d
from distantspeech_amd import _lib as L
C, N, K, B = 4, 20, 129, 256
rng = np.random.default_rng(0)
for T in (1, 50):
    D = ((rng.standard_normal((B, T, K, C)) + 1j * rng.standard_normal((B, T, K, C))) * 0.3).astype(np.complex64)
    for fp64 in (0, 1):
        e = d.BatchEngine(L.ALGO_WPE, C, 256, batch=B, filter_len=N, rls_lambda=0.998)
        if fp64: e.set_param_i(L.PARAM_WPE_FP64, 1)
        e.wpe_update(D, D)
        t0 = time.perf_counter(); n = 3
        for _ in range(n): e.wpe_update(D, D)
        dt = (time.perf_counter() - t0) / n
        print("B=%d T=%d fp64=%d: %.2f ms per call incl. host copies = %.0f k frames/s" % (B, T, fp64, dt * 1e3, B * T / dt / 1e3))
PY
