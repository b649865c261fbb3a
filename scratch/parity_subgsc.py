"""Where does the SubbandGSC (G12) error enter?  CPU-only: the chain's stages through tests/emul, composed as ds_api_chains.hip::chain2_run
composes them, with the option of replacing single stages' outputs by the fp64 oracle's to bisect."""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import load, as_float, rms
from emul import emul as E
from emul.emul import EmulOp, EmulTransform, EmulFrontend
from oracle import ds_oracle as O

vp = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
f32 = ctypes.c_float


def emul_chain(x, M, FL, coef, Fn, rls, p_override=None, T_call=None):
    """x [M, L] float32 -> (output, bm [L, M], p [K, T], aligned [L, M])"""
    lib = E.lib()
    nfft, K = 2 * FL, FL + 1
    L = x.shape[1]; T = L // FL
    fe = EmulFrontend(M, coef=coef, radius=0.98)
    xn = fe.dcnotch(x[None])                                   # [1, M, L]
    xa, fixed = fe.firbank(np.ascontiguousarray(np.swapaxes(xn, 1, 2)))      # [1, L, M], [1, L]
    D = EmulTransform(nfft, M).stft(xa, 0)                     # [1, T, K, M]
    sp = EmulOp("mcspp", nfft, M=M)
    p = np.concatenate([sp.run_mcspp(D[:, :5], Fn, variant=12)[0], sp.run_mcspp(D[:, 5:], Fn, variant=13)[0]], axis=1)   # [1, T, K]
    if p_override is not None:
        p = np.ascontiguousarray(p_override.T[None], dtype=np.float32)
    F = EmulTransform(nfft, 1).stft(fixed[:, :, None], 0)[..., 0]            # [1, T, K]
    N = 2
    KP = (K + 3) & ~3
    NF = 4 * N + 2 * N * N if rls else 4 * N + 1
    st = np.zeros((M, NF, KP), dtype=np.float32)
    if rls:
        for i in range(N):
            st[:, 4 * N + 2 * (i * N + i), :] = 1000.0
    e = np.zeros((M, T, K), dtype=np.complex64)
    rc = lib.emul_fan(4 if rls else 3, 1, M, M, K, T, vp(st), NF, vp(np.ascontiguousarray(F)), vp(np.ascontiguousarray(D)), None if rls else vp(p), vp(e),
                      int(not rls), 1, f32(0.5 if rls else 0.1), f32(0.9), f32(1e-4), f32(0.998))
    assert rc == 0
    bm = EmulTransform(nfft, 1, batch=M).istft(e[..., None])[:, :, 0]         # [M, L]
    Xaic = EmulTransform(nfft, M).stft(np.ascontiguousarray(bm.T)[None], 0)   # [1, T, K, M]
    aic = EmulOp("sublms", nfft, M=M, N=2, mu=0.01, alpha=0.8)
    Fd = np.concatenate([np.zeros_like(F[:, :1]), F[:, :-1]], axis=1)         # delay_fbf in the spectral domain
    e2 = aic.run(Xaic, np.ascontiguousarray(Fd), np.ascontiguousarray((np.float32(1) - p).astype(np.float32)), out_complex=True)[0]
    out = EmulTransform(nfft, 1).istft(e2[..., None])[0, :, 0]
    return out, bm.T, p[0].T, xa[0]


for name in ("rec1", "synth_m6", "synth_m6_rls"):
    g = load("g12_subbandgsc_" + name)
    M, FL, rls = [int(v) for v in g["params"]]
    x = as_float(g["x"]).astype(np.float32)
    mic = O.OracleMicArray(arrayType="circular", r=float(g["r"]), M=M, n_fft=512)
    coef = np.ascontiguousarray(g["delay_filter"], dtype=np.float32)
    if coef.shape[0] == M: coef = np.ascontiguousarray(coef.T)
    Fn = O.gen_noise_msc(O.OracleMicArray(arrayType="circular", r=0.032, M=M), 512)[:, 1, 2]
    out, bm, p, al = emul_chain(x, M, FL, coef, Fn, rls)
    print(name, "out rms err %.3e (ref rms %.3e)  bm %.3e (ref %.3e)  aligned %.3e  p: median %.2e max %.2e" % (
        rms(out - g["output"]), rms(g["output"]), rms(bm - g["bm_output"]), rms(g["bm_output"]), rms(al - g["aligned_output"]),
        np.median(np.abs(p - g["p"])), np.abs(p - g["p"]).max()))
    out2, bm2, _, _ = emul_chain(x, M, FL, coef, Fn, rls, p_override=g["p"])
    print("   with the reference's p:  out %.3e  bm %.3e" % (rms(out2 - g["output"]), rms(bm2 - g["bm_output"])))
