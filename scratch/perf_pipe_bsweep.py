"""hop period of the frame / pipelined kernels vs workgroups per CU (B = 256 .. 1024 = 1 .. 4 per CU), 625 hops per call"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from distantspeech_amd import BatchEngine, _lib as L
M, NFFT, HOP, T = 4, 512, 256, 625
dev = torch.device("cuda", 0)
omega = 2 * np.pi * np.arange(257) * 16000 / 512
tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
steer = np.exp(-1j * omega[:, None] * tao[None, :])
for name, algo in (("MVDR", L.ALGO_ADAPTIVE), ("fixed", L.ALGO_FIXED), ("GSC", L.ALGO_GSC)):
    for B in (256, 512, 768, 1024, 2048):
        Ltot = T * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        out = []
        for tag, min_t in (("frame", 1 << 30), ("pipe", 1)):
            os.environ["DS_PIPE_MIN_T"] = str(min_t)
            e = BatchEngine(algo, M, NFFT, HOP, batch=B, device=0)
            e.set_steering(steer / M if algo == L.ALGO_FIXED else steer)
            e.set_split(1)
            if algo != L.ALGO_FIXED:
                e.set_method(2)
            best = 1e9
            for rnd in range(3):
                e.synchronize(); e.timing_begin()
                e.process_device_seq(x.data_ptr(), 1, M * Ltot, Ltot, T * HOP, T * HOP, 1, y.data_ptr(), Ltot, T * HOP, graph=0)
                best = min(best, e.timing_end())
            out.append("%s %.3f ms (%.2f us/hop, %.1f M fr/s)" % (tag, best, best / T * 1e3, B * T / best / 1e3))
            e.close()
        print("%-6s B=%4d  %s" % (name, B, " | ".join(out)), flush=True)
        del x, y
