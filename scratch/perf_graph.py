"""T=1 launch time: plain stream launches vs hipGraph replay of the same sequence."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distantspeech_amd import BatchEngine, _lib as L
M, NFFT, HOP = 4, 512, 256
dev = torch.device("cuda", 0)
for B in (256, 1024, 4096):
    K = 400; Ltot = K * HOP
    x = torch.randn((B, M, Ltot), device=dev) * 0.05
    y = torch.empty((B, Ltot), device=dev)
    eng = BatchEngine(1, M, NFFT, HOP, batch=B, device=0)
    eng.set_steering(np.ones((257, 4), np.complex64)); eng.set_method(2)
    torch.cuda.synchronize()
    xp, yp = x.data_ptr(), y.data_ptr()
    for graph in (0, 1):
        best = 1e9
        for _ in range(6):
            eng.synchronize(); eng.timing_begin()
            eng.process_device_seq(xp, 1, M * Ltot, Ltot, HOP, HOP, K, yp, Ltot, HOP, graph=graph)
            best = min(best, eng.timing_end())
        print("B=%d graph=%d: %.2f us/launch  %.1f M frames/s" % (B, graph, best / K * 1e3, B * K / best / 1e3), flush=True)
    del x, y, eng
