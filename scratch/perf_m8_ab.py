"""8-microphone adaptive MVDR frame kernel: quad-spread per-bin program (DS_M8_QUAD=1) vs one thread per bin (default), B = 1024,
white-noise input, one hop per call and 40 hops per call.  python scratch/perf_m8_ab.py [nfft ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from distantspeech_amd import BatchEngine
    M, NFFT = 8, int(sys.argv[2])
    dev = torch.device("cuda", 0)
    out = []
    for T in (1, 40):
        HOP, B = NFFT // 2, 1024
        K = 80 // T; Ltot = (K + 2) * T * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(1, M, NFFT, HOP, batch=B, device=0)
        eng.set_steering(np.ones((NFFT // 2 + 1, M), np.complex64)); eng.set_method(2)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            eng.synchronize(); eng.timing_begin()
            eng.process_device_seq(x.data_ptr(), 1, M * Ltot, Ltot, T * HOP, T * HOP, K, y.data_ptr(), Ltot, T * HOP, graph=0)
            best = min(best, eng.timing_end())
        out.append("T=%2d %8.2f us/launch %6.2f M frames/s" % (T, best / K * 1e3, B * T * K / best / 1e3))
        del x, y, eng
    print(" | ".join(out)); sys.exit(0)
for nfft in (sys.argv[1:] or ["512", "1024"]):
    for name, env in (("quad", {"DS_M8_QUAD": "1"}), ("one-thread", {})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", nfft], env=dict(os.environ, **env), capture_output=True, text=True)
        print("M=8 nfft=%4s %-10s %s" % (nfft, name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
