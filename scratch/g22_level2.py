"""Which operator moves p at ten times the level?  McSpp (emulated kernel program, fp32) against OracleMcSpp (fp64) on the SAME spectra D
(the oracle's), and its parts: Gamma (McCDR), xi, gamma."""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import load, rms
from emul.emul import EmulOp
from oracle import ds_oracle as O

g = load("g22_subbandgsc_pf_rec1_1")
x16 = g["x"]; M, FL = 4, 256; nfft = 512; K = 257
g12 = load("g12_subbandgsc_rec1")
mic = O.OracleMicArray(arrayType="circular", r=0.032, M=M, n_fft=512)
Fn = O.gen_noise_msc(O.OracleMicArray(arrayType="circular", r=0.032, M=M), nfft)[:, 1, 2]
for scale in (1.0, 10.0):
    x = (x16.astype(np.float32) / 32768.0 * np.float32(scale)).astype(np.float64)
    o = O.OracleSubbandGSC(mic, frameLen=FL)
    al = o.process(x)[4]                                   # aligned [L, M]
    D = O.OracleTransform(channel=M, n_fft=nfft, hop_length=FL).stft(al)      # [K, T, M]
    T = D.shape[1]
    est = O.OracleMcSpp(nfft=nfft, channels=M)
    P = np.zeros((T, K)); XI = np.zeros((T, K)); GA = np.zeros((T, K)); Q = np.zeros((T, K))
    for n in range(T):
        est.estimation(D[:, n, :]); P[n] = est.p; XI[n] = est.xi; GA[n] = est.gamma
    Dk = np.ascontiguousarray(np.transpose(D, (1, 0, 2))[None]).astype(np.complex64)      # [1, T, K, M]
    sp = EmulOp("mcspp", nfft, M=M)
    pk = np.concatenate([sp.run_mcspp(Dk[:, :5], Fn, variant=12)[0], sp.run_mcspp(Dk[:, 5:], Fn, variant=13)[0]], axis=1)[0]
    dp = np.abs(pk - P)
    t, k = np.unravel_index(np.argmax(dp), dp.shape)
    print("scale %g: p max diff %.3e at frame %d bin %d (p ref %.4f, xi %.3e, gamma %.3e); count > 1e-3: %d of %d; median %.1e" % (scale, dp.max(), t, k, P[t, k], XI[t, k], GA[t, k], (dp > 1e-3).sum(), dp.size, np.median(dp)))
    big = np.argwhere(dp > 1e-3)
    print("   bins of the large ones:", sorted(set(big[:, 1].tolist()))[:30], "frames:", sorted(set(big[:, 0].tolist()))[:30])

print("--- sensitivity of the fp64 McSpp itself to a 1e-5 / 1e-6 relative perturbation of its input spectra")
rng = np.random.default_rng(0)
for scale in (1.0, 10.0):
    x = (x16.astype(np.float32) / 32768.0 * np.float32(scale)).astype(np.float64)
    o = O.OracleSubbandGSC(mic, frameLen=FL)
    al = o.process(x)[4]
    for rel in (1e-5, 1e-6):
        al2 = al + rel * rms(al) * rng.standard_normal(al.shape)
        res = []
        for a in (al, al2):
            D = O.OracleTransform(channel=M, n_fft=nfft, hop_length=FL).stft(a)
            est = O.OracleMcSpp(nfft=nfft, channels=M)
            P = np.zeros((D.shape[1], K))
            for n in range(D.shape[1]):
                est.estimation(D[:, n, :]); P[n] = est.p
            res.append(P)
        d = np.abs(res[0] - res[1])
        print("scale %g, input perturbed by %.0e: p max diff %.3e, count > 1e-3: %d" % (scale, rel, d.max(), (d > 1e-3).sum()))
