"""PCIe-inclusive rate of the host-pointer entry point (ds_process: H2D of the hop, kernel, D2H of the output, synchronous) for the headline
workload, B = 1024, one hop and 125 hops per call.  usage: python scratch/perf_host_path.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import distantspeech_amd as ds
from distantspeech_amd import _lib as L
B, M, nfft, hop = 1024, 4, 512, 256
e = ds.BatchEngine(L.ALGO_ADAPTIVE, M, nfft, hop, batch=B, device=0)
e.set_steering(np.ones((nfft // 2 + 1, M), np.complex64)); e.set_method(L.METHOD_MVDR)
rng = np.random.default_rng(0)
for T in (1, 8, 125):
    x = (rng.standard_normal((B, M, T * hop)) * 0.05).astype(np.float32)
    for _ in range(3): e.process(x, L.LAYOUT_CHANNELS_SAMPLES)
    n = max(3, 200 // T)
    t0 = time.perf_counter()
    for _ in range(n): e.process(x, L.LAYOUT_CHANNELS_SAMPLES)
    dt = time.perf_counter() - t0
    print("ds_process host path, B=%d, %d hop(s) per call: %.2f M frames/s (%.1f us per call, %.1f MB in + %.1f MB out per call)" %
          (B, T, B * T * n / dt / 1e6, dt / n * 1e6, x.nbytes / 1e6, B * T * hop * 4 / 1e6), flush=True)
