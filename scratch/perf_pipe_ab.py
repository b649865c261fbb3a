"""A/B of the hop-pipelined frame kernel (ds_pipe.hpp) against the phase-by-phase one on one box: python scratch/perf_pipe_ab.py
DS_PIPE_MIN_T is read at ds_create, so both kernels run in ONE process, alternately, on the same inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from distantspeech_amd import BatchEngine, _lib as L

M, NFFT, HOP = 4, 512, 256
dev = torch.device("cuda", 0)
omega = 2 * np.pi * np.arange(257) * 16000 / 512
tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
steer = np.exp(-1j * omega[:, None] * tao[None, :])
configs = [("cfg2 MVDR", L.ALGO_ADAPTIVE, 1024), ("cfg3 GSC", L.ALGO_GSC, 4096), ("fixed", L.ALGO_FIXED, 1024)]
if len(sys.argv) > 1:
    configs = [c for c in configs if c[0].split()[0] in sys.argv[1:]]
for name, algo, B in configs:
    for T in (625, 64, 8):
        n = max(1, 1250 // T)
        Ltot = n * T * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        res = {}
        engs = {}
        for tag, min_t in (("frame", 1 << 30), ("pipe", 1)):
            os.environ["DS_PIPE_MIN_T"] = str(min_t)
            e = BatchEngine(algo, M, NFFT, HOP, batch=B, device=0)
            e.set_steering(steer / M if algo == L.ALGO_FIXED else steer)
            if algo != L.ALGO_FIXED:
                e.set_method(2)
            engs[tag] = e
        torch.cuda.synchronize()
        for rnd in range(4):
            for tag, e in engs.items():
                e.synchronize(); e.timing_begin()
                e.process_device_seq(x.data_ptr(), 1, M * Ltot, Ltot, T * HOP, T * HOP, n, y.data_ptr(), Ltot, T * HOP, graph=0)
                ms = e.timing_end()
                res[tag] = min(res.get(tag, 1e9), ms)
        print("%-10s B=%d T=%3d  frame %8.3f ms = %6.1f M frames/s | pipe %8.3f ms = %6.1f M frames/s | x%.3f" % (
            name, B, T, res["frame"], B * n * T / res["frame"] / 1e3, res["pipe"], B * n * T / res["pipe"] / 1e3, res["frame"] / res["pipe"]), flush=True)
        for e in engs.values():
            e.close()
        del x, y
        torch.cuda.empty_cache()
