import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distantspeech_amd import BatchEngine, _lib as L
M, NFFT, HOP = 4, 512, 256
dev = torch.device("cuda", 0)
for algo, name in ((L.ALGO_ADAPTIVE, "adaptive"), (L.ALGO_GSC, "gsc")):
    for B in (1024, 4096):
        K = 400; Ltot = K * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        yref = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(algo, M, NFFT, HOP, batch=B, device=0)
        omega = 2 * np.pi * np.arange(257) * 16000 / 512
        tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
        eng.set_steering(np.exp(-1j * omega[:, None] * tao[None, :])); eng.set_method(2)
        torch.cuda.synchronize()
        xp, yp = x.data_ptr(), y.data_ptr()
        for split in (1, 2, 4, 8):
            eng.reset(); eng.set_split(split)
            eng.process_device_seq(xp, 1, M * Ltot, Ltot, HOP, HOP, K, yp, Ltot, HOP, graph=2)
            best = 1e9
            for _ in range(4):
                eng.reset()
                eng.synchronize(); eng.timing_begin()
                eng.process_device_seq(xp, 1, M * Ltot, Ltot, HOP, HOP, K, yp, Ltot, HOP, graph=1)
                best = min(best, eng.timing_end())
            if split == 1: yref.copy_(y)
            same = bool(torch.equal(y, yref))
            print("%s B %d split %d: %.2f us/step %.1f Mframes/s  same_output=%s" % (name, B, split, best / K * 1e3, B * K / best / 1e3, same), flush=True)
        del x, y, eng
