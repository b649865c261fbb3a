import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import load, rms
from emul.emul import EmulFrontend
from oracle import ds_oracle as O
g = load("g22_subbandgsc_pf_rec1_1"); g12 = load("g12_subbandgsc_rec1")
x16 = g["x"]; M = 4
coef = np.ascontiguousarray(g12["delay_filter"], dtype=np.float32)
x = (x16.astype(np.float32) / 32768.0).astype(np.float32)
fe = EmulFrontend(M, coef=coef, radius=0.98)
xn = fe.dcnotch(x[None])[0]                                  # [M, L]
ref = np.stack([O.OracleDcNotch(radius=0.98).filter(x[m].astype(np.float64)) for m in range(M)])
print("notch: rel err %.3e" % (rms(xn - ref) / rms(ref)))
xa, fixed = fe.firbank(np.ascontiguousarray(xn.T)[None])
# FIR on the exact notch output (oracle) in double
mic = O.OracleMicArray(arrayType="circular", r=0.032, M=M, n_fft=512)
ta = O.OracleTimeAlignment(mic, np.array([197, 0]) / 180 * np.pi)
ral = ta.process(ref.T.copy()) if hasattr(ta, "process") else None
print("aligned: rel err %.3e" % (rms(xa[0] - ral) / rms(ral)))
# FIR alone: feed the emulated FIR with the oracle's notch output rounded to fp32
fe2 = EmulFrontend(M, coef=coef, radius=0.98)
xa2, _ = fe2.firbank(np.ascontiguousarray(ref.T.astype(np.float32))[None])
print("FIR alone (exact notch in, rounded to fp32): rel err %.3e" % (rms(xa2[0] - ral) / rms(ral)))
