"""Where does the notebook-MVDR error enter?  CPU-only: the kernel's per-bin program through tests/emul against the G11 fixtures."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import load, as_float, rms
from emul.emul import EmulOp, EmulTransform
from oracle import ds_oracle as O

for name in ("rec1", "synth_m6"):
    g = load("g11_mcspp_" + name)
    M, nfft, hop = [int(v) for v in g["params"]]
    x = as_float(g["x"])
    tf = EmulTransform(nfft, M)
    D = tf.stft(np.ascontiguousarray(x.T)[None], 0)
    Fn = O.gen_noise_msc(O.OracleMicArray(arrayType="circular", r=0.032, M=M), nfft)[:, 1, 2]
    op = EmulOp("mcspp", nfft, M=M)
    p, w, yout, pxx, pinv = op.run_mcspp(D, Fn, want_matrices=True)
    y = tf.istft(np.ascontiguousarray(yout[..., None]))[0, :, 0]
    Yref = g["Yout"]
    T = Yref.shape[0]
    e = np.abs(yout[0] - Yref)
    print(name, "rms(y-ref) %.3e  rms(ref) %.3e  rel %.3e" % (rms(y - g["y"]), rms(g["y"]), rms(y - g["y"]) / rms(g["y"])))
    print("  per-frame rms err:", " ".join("%.1e" % v for v in np.sqrt((e ** 2).mean(axis=1))[:: max(1, T // 24)]))
    print("  per-frame rms ref:", " ".join("%.1e" % v for v in np.sqrt((np.abs(Yref) ** 2).mean(axis=1))[:: max(1, T // 24)]))
    eb = np.sqrt((e[12:] ** 2).mean(axis=0))
    print("  worst bins (frames >= 12):", np.argsort(eb)[-8:], eb[np.argsort(eb)[-8:]])
    print("  p err: median %.2e max %.2e" % (np.median(np.abs(p[0] - g["p"])), np.abs(p[0] - g["p"]).max()))
    # oracle in double from the SAME complex64 spectra: isolates the operator from the transform
    est = O.OracleMcSpp(nfft=nfft, channels=M)
    Yo = np.zeros_like(Yref, dtype=complex)
    for n in range(T):
        est.estimation(D[0, n].astype(complex))
        sv = O.steering(est.Phi_xx)
        wv = O.compute_mvdr_weight(sv, est.Phi_vv_inv)
        Yo[n] = np.einsum("ij,ij->i", wv.conj(), D[0, n])
    print("  oracle(fp64, emul spectra) vs fixture: %.3e ; emul vs that oracle %.3e" % (rms(Yo - Yref), rms(yout[0] - Yo)))
