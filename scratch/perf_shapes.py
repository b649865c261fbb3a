"""T=1 and chunked rates of the fused frame kernels for every (M, nfft) shape of the BASELINE configs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distantspeech_amd import BatchEngine, _lib as L
dev = torch.device("cuda", 0)
for algo, name in ((1, "mvdr"), (2, "gsc")):
    for M, NFFT in ((4, 256), (4, 512), (6, 512), (4, 1024), (6, 1024), (8, 1024)):
        HOP, B = NFFT // 2, 1024
        for T in (1, 40):
            K = 80 // T; Ltot = (K + 2) * T * HOP
            x = torch.randn((B, M, Ltot), device=dev) * 0.05
            y = torch.empty((B, Ltot), device=dev)
            eng = BatchEngine(algo, M, NFFT, HOP, batch=B, device=0)
            eng.set_steering(np.ones((NFFT // 2 + 1, M), np.complex64)); eng.set_method(2)
            torch.cuda.synchronize()
            xp, yp = x.data_ptr(), y.data_ptr()
            best = 1e9
            for _ in range(4):
                eng.synchronize(); eng.timing_begin()
                eng.process_device_seq(xp, 1, M * Ltot, Ltot, T * HOP, T * HOP, K, yp, Ltot, T * HOP, graph=0)
                best = min(best, eng.timing_end())
            print("%-4s M=%d nfft=%4d T=%2d: %8.2f us/launch %7.2f M frames/s" % (name, M, NFFT, T, best / K * 1e3, B * K * T / best / 1e3), flush=True)
            del x, y, eng
