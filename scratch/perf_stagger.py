"""T=1 launch time vs intra-launch stagger settings (DS_STAGGER=naps,bit)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from distantspeech_amd import BatchEngine, _lib as L
    M, NFFT, HOP = 4, 512, 256
    algo = int(os.environ.get("AB_ALGO", "1"))
    dev = torch.device("cuda", 0)
    out = []
    for B in (1024, 2048):
        K = 400; Ltot = K * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(algo, M, NFFT, HOP, batch=B, device=0)
        omega = 2 * np.pi * np.arange(257) * 16000 / 512
        tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
        eng.set_steering(np.exp(-1j * omega[:, None] * tao[None, :])); eng.set_method(2)
        torch.cuda.synchronize()
        xp, yp = x.data_ptr(), y.data_ptr()
        best = 1e9
        for _ in range(6):
            eng.synchronize(); eng.timing_begin()
            eng.process_device_seq(xp, 1, M * Ltot, Ltot, HOP, HOP, K, yp, Ltot, HOP, graph=0)
            best = min(best, eng.timing_end())
        out.append("B%d %.2fus %.1fM" % (B, best / K * 1e3, B * K / best / 1e3))
        del x, y, eng
    print(" | ".join(out))
    sys.exit(0)
settings = ["0,0"] + ["%d,%d" % (n, b) for b in (0, 1, 5, 8, 9) for n in (3, 5, 7, 10)]
for s in settings:
    env = dict(os.environ, DS_STAGGER=s)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print("%-8s %s" % (s, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
