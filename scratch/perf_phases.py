"""Per-phase timeline of workgroup 0 of the MVDR frame kernel (phase-timing variant build)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DSENH_LIB"] = os.path.join(ROOT, "scratch", "variants", "libdsenh_phase0.so")
sys.path.insert(0, ROOT)
from distantspeech_amd import BatchEngine, _lib as L
M, NFFT, HOP = 4, 512, 256
dev = torch.device("cuda", 0)
lib = L.load()
lib.ds_debug_phase_times.restype = ctypes.c_int
for B in (64, 256, 1024, 4096):
    K = 50; Ltot = K * HOP
    x = torch.randn((B, M, Ltot), device=dev) * 0.05
    y = torch.empty((B, Ltot), device=dev)
    eng = BatchEngine(1, M, NFFT, HOP, batch=B, device=0)
    omega = 2 * np.pi * np.arange(257) * 16000 / 512
    tao = -0.032 * np.cos(3.438 - np.arange(4) * np.pi / 2) / 343
    eng.set_steering(np.exp(-1j * omega[:, None] * tao[None, :])); eng.set_method(2)
    torch.cuda.synchronize()
    xp, yp = x.data_ptr(), y.data_ptr()
    eng.process_device_seq(xp, 1, M * Ltot, Ltot, HOP, HOP, K, yp, Ltot, HOP, graph=0)
    eng.synchronize()
    wb = (ctypes.c_ulonglong * (2 * B))()
    lib.ds_debug_wg_times(wb, B)
    w = np.array(wb[:], dtype=np.float64).reshape(B, 2) * 10.0
    t0 = w[:, 0].min()
    st, en = (w[:, 0] - t0) / 1e3, (w[:, 1] - t0) / 1e3
    q = lambda a: " ".join("%.2f" % v for v in np.percentile(a, [0, 10, 50, 90, 100]))
    late = (st > 3.0).mean()
    order = np.argsort(st)
    print("B=%d late starters (>3us): %.1f%%; first late block ids: %s" % (B, 100 * late, np.sort(np.nonzero(st > 3.0)[0])[:12]))
    print("   workgroup start us (min p10 p50 p90 max): %s | end: %s | duration: %s" % (q(st), q(en), q(en - st)), flush=True)
    del x, y, eng
