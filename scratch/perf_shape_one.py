"""One (algo, M, nfft) shape of scratch/perf_shapes.py under each named variant library: python scratch/perf_shape_one.py M nfft v1 v2 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "--child":
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    from distantspeech_amd import BatchEngine
    M, NFFT = int(sys.argv[2]), int(sys.argv[3])
    dev = torch.device("cuda", 0)
    out = []
    for T in (1, 40):
        HOP, B = NFFT // 2, int(os.environ.get('DS_SHAPE_B', '1024'))
        K = 80 // T; Ltot = (K + 2) * T * HOP
        x = torch.randn((B, M, Ltot), device=dev) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(int(os.environ.get('DS_SHAPE_ALGO', '1')), M, NFFT, HOP, batch=B, device=0)
        eng.set_steering(np.ones((NFFT // 2 + 1, M), np.complex64)); eng.set_method(2)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            eng.synchronize(); eng.timing_begin()
            eng.process_device_seq(x.data_ptr(), 1, M * Ltot, Ltot, T * HOP, T * HOP, K, y.data_ptr(), Ltot, T * HOP, graph=0)
            best = min(best, eng.timing_end())
        out.append("T=%d %.2f us" % (T, best / K * 1e3))
        del x, y, eng
    print(" | ".join(out)); sys.exit(0)
M, NFFT = sys.argv[1], sys.argv[2]
for n in sys.argv[3:]:
    env = dict(os.environ, DSENH_LIB=os.path.join(ROOT, "scratch", "variants", "libdsenh_%s.so" % n))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", M, NFFT], env=env, capture_output=True, text=True)
    print("%-10s %s" % (n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
