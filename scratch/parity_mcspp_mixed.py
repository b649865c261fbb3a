"""Which precision does the notebook MVDR need?  fp64 oracle with (a) state rounded to fp32 after every frame, (b) everything fp64 but eigh in fp32 ..."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import load, as_float, rms
from emul.emul import EmulTransform
from oracle import ds_oracle as O

def c64(a): return a.astype(np.complex64).astype(complex)
def f32(a): return a.astype(np.float32).astype(float)

for name in ("rec1", "synth_m6"):
    g = load("g11_mcspp_" + name)
    M, nfft, hop = [int(v) for v in g["params"]]
    x = as_float(g["x"])
    tf = EmulTransform(nfft, M)
    D = tf.stft(np.ascontiguousarray(x.T)[None], 0)
    Yref = g["Yout"]; T = Yref.shape[0]
    for mode in ("fp64", "state32", "state32+xx"):
        est = O.OracleMcSpp(nfft=nfft, channels=M)
        Yo = np.zeros_like(Yref, dtype=complex)
        Pxx = np.zeros_like(est.Phi_yy)
        for n in range(T):
            y = D[0, n].astype(complex)
            if mode == "state32+xx":
                # Phi_xx carried by its own recursion: Phi_xx' = 0.92 Phi_xx + 0.08 p (psd - Phi_vv_sym)  (p of the previous frame's update)
                pass
            est.estimation(y)
            if mode != "fp64":
                est.Phi_yy = c64(est.Phi_yy); est.Phi_vv = c64(est.Phi_vv)
                est.mccdr.Pxii = f32(est.mccdr.Pxii); est.mccdr.Pxij12 = c64(est.mccdr.Pxij12)
            sv = O.steering(est.Phi_xx)
            wv = O.compute_mvdr_weight(sv, est.Phi_vv_inv)
            Yo[n] = np.einsum("ij,ij->i", wv.conj(), y)
        e = np.abs(Yo - Yref)
        y_t = tf.__class__(nfft, 1).istft(np.ascontiguousarray(Yo.astype(np.complex64)[None, :, :, None]))[0, :, 0]
        print(name, mode, "spec rms err %.3e  time rms err %.3e" % (rms(Yo - Yref), rms(y_t - g["y"])), " worst bins", np.argsort((e**2).mean(0))[-4:])
