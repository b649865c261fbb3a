"""A/B timing of kernel builds on one GPU box: python scratch/perf_ab.py name1 name2 ... (libs in scratch/variants/)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:]
if names and names[0] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from distantspeech_amd import BatchEngine, _lib as L
    M, NFFT, HOP = 4, 512, 256
    algo = int(os.environ.get("AB_ALGO", "1"))
    dev = torch.device("cuda", 0)
    out = []
    for B in (1024, 4096):
        K = 400; Ltot = K * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(algo, M, NFFT, HOP, batch=B, device=0)
        omega = 2 * np.pi * np.arange(257) * 16000 / 512
        tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
        eng.set_steering(np.exp(-1j * omega[:, None] * tao[None, :])); eng.set_method(2)
        torch.cuda.synchronize()
        xp, yp = x.data_ptr(), y.data_ptr()
        for T in (1, 400):
            n = K // T
            best = 1e9
            for _ in range(5):
                eng.synchronize(); eng.timing_begin()
                eng.process_device_seq(xp, 1, M * Ltot, Ltot, T * HOP, T * HOP, n, yp, Ltot, T * HOP, graph=0)
                best = min(best, eng.timing_end())
            out.append("B%d T%d %.2fus %.1fM" % (B, T, best / n * 1e3, B * n * T / best / 1e3))
        del x, y, eng
    print(" | ".join(out))
    sys.exit(0)
for rnd in range(2):
    for n in names:
        env = dict(os.environ, DSENH_LIB=os.path.join(ROOT, "scratch", "variants", "libdsenh_%s.so" % n))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        print("%-12s %s" % (n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
