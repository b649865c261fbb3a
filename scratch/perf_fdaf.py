"""FDAF kernel timing: B instances x T blocks, device-resident I/O, HIP events on the engine stream."""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from distantspeech_amd import _lib as L
from distantspeech_amd.engine import BatchEngine

def run(B, C, Lf, T, kind, trunc, pmode, reps=5):
    eng = BatchEngine(L.ALGO_FDAF, C, 2 * Lf, batch=B, filt_mu=0.05, filt_alpha=0.9)
    eng.set_fdaf(kind, constrain=True, non_causal=(kind == 0), weight_norm=(kind == 2))
    dev = torch.device("cuda:0")
    x = (torch.randn(B, T * Lf, C, device=dev) * 0.1).contiguous()
    d = (torch.randn(B, T * Lf, device=dev) * 0.1).contiguous()
    K = Lf + 1
    p = torch.rand(B, T, K, device=dev).contiguous() if pmode == 2 else None
    e = torch.empty_like(d)
    torch.cuda.synchronize()
    lib = eng._lib
    vp = ctypes.c_void_p
    def call():
        L.check(lib.ds_fdaf_update(eng._h, vp(x.data_ptr()), vp(d.data_ptr()), vp(p.data_ptr()) if p is not None else None,
                                   pmode, T, trunc, vp(e.data_ptr()), None, L.MEM_DEVICE), eng._h)
    call(); lib.ds_synchronize(eng._h)
    ms = ctypes.c_float()
    best = 1e9
    for _ in range(reps):
        lib.ds_timing_begin(eng._h); call(); lib.ds_timing_end(eng._h, ctypes.byref(ms)); best = min(best, ms.value)
    blocks = B * T
    print("B=%5d C=%d L=%3d T=%3d kind=%d trunc=%3d: %8.3f ms  %7.2f M blocks/s  %6.2f us/block/WG-serial  (%.1f M samples/s)" %
          (B, C, Lf, T, kind, trunc, best, blocks / best / 1e3, best * 1e3 / T, blocks * Lf / best / 1e3))
    eng.close()

for B in (256, 1024, 4096):
    run(B, 3, 256, 20, 0, 30, 2)       # TDGSC canceller (M=4)
run(4096, 1, 256, 20, 1, -1, 0)        # FDGSC blocking filters (B*M instances)
run(1024, 4, 256, 20, 2, -1, 0)        # FDGSC canceller
run(1024, 7, 256, 20, 0, 30, 2)        # TDGSC canceller (M=8)
run(1024, 1, 64, 20, 0, -1, 0)
run(1024, 1, 512, 20, 0, -1, 0)
