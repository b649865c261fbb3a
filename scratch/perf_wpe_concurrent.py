"""Is one WPE launch over 1024 utterances slower than S concurrent launches over 1024 / S on S streams?  (where the +10 % of two
free-running half-batch chains comes from).  usage: python scratch/perf_wpe_concurrent.py"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import distantspeech_amd as ds
from distantspeech_amd import _lib as L
from _cases import DeviceBuffers

B, M, nfft, hop, T, N = 1024, 8, 1024, 512, 1, 40
K = nfft // 2 + 1
dv = DeviceBuffers()
rng = np.random.default_rng(0)
z = (rng.standard_normal((B, T, K, M, 2)) * 0.05).astype(np.float32)
xd = dv.upload(z); dd = dv.upload(z[::-1].copy()); ed = dv.zeros(z.nbytes)
lib = L.load()
per = T * K * M * 8
for S in (1, 2, 4, 1, 2):
    hs = [ds.BatchEngine(L.ALGO_WPE, M, nfft, hop, batch=B // S, device=0, filter_len=2) for _ in range(S)]
    def run(n):
        for _ in range(n):
            for s, e in enumerate(hs):
                o = s * (B // S) * per
                L.check(lib.ds_wpe_update(e._h, ctypes.c_void_p(xd + o), ctypes.c_void_p(dd + o), T, ctypes.c_void_p(ed + o), L.MEM_DEVICE), e._h)
        for e in hs: e.synchronize()
    run(5)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); run(N); best = min(best, time.perf_counter() - t0)
    print(f"WPE S={S}: {best / N * 1e6:7.1f} us per {B}-utterance frame", flush=True)
    for e in hs: e.close()
dv.free()
