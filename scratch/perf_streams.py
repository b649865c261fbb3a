import os, sys, numpy as np, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distantspeech_amd import BatchEngine, _lib as L
M, NFFT, HOP = 4, 512, 256
dev = torch.device("cuda", 0)
B, K = 1024, 400
Ltot = K * HOP
x = torch.randn((B, M, Ltot), device=dev) * 0.05
y = torch.empty((B, Ltot), device=dev)
eng = BatchEngine(L.ALGO_ADAPTIVE, M, NFFT, HOP, batch=B, device=0)
omega = 2 * np.pi * np.arange(257) * 16000 / 512
tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
eng.set_steering(np.exp(-1j * omega[:, None] * tao[None, :])); eng.set_method(2)
torch.cuda.synchronize()
xp, yp = x.data_ptr(), y.data_ptr()
for ns in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    per = B // ns
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(ns):
            first = s * per
            eng.process_device_seq(xp + 4 * first * M * Ltot, 1, M * Ltot, Ltot, HOP, HOP, K, yp + 4 * first * Ltot, Ltot, HOP,
                                   first=first, count=per, stream=streams[s].cuda_stream)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("streams %d: %.2f us/step  %.1f Mframes/s" % (ns, best / K * 1e6, B * K / best / 1e6), flush=True)
