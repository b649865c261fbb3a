"""T=1 launch time vs batch size (how much of the launch is per-workgroup latency chain vs throughput)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distantspeech_amd import BatchEngine, _lib as L
M, NFFT, HOP = 4, 512, 256
dev = torch.device("cuda", 0)
for algo in (1, 0):
    for B in (64, 256, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096, 8192):
        K = 200; Ltot = K * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(algo, M, NFFT, HOP, batch=B, device=0)
        omega = 2 * np.pi * np.arange(257) * 16000 / 512
        tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
        eng.set_steering(np.exp(-1j * omega[:, None] * tao[None, :]))
        if algo == 1: eng.set_method(2)
        torch.cuda.synchronize()
        xp, yp = x.data_ptr(), y.data_ptr()
        best = 1e9
        for _ in range(5):
            eng.synchronize(); eng.timing_begin()
            eng.process_device_seq(xp, 1, M * Ltot, Ltot, HOP, HOP, K, yp, Ltot, HOP, graph=0)
            best = min(best, eng.timing_end())
        print("algo %d B %5d  %7.2f us/launch  %6.1f M frames/s" % (algo, B, best / K * 1e3, B * K / best / 1e3), flush=True)
        del x, y, eng
