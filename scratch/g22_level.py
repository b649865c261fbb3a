"""Which stage of the SubbandGSC chain loses two decades of relative accuracy when rec1 runs at ten times its level?
CPU only: the chain's kernel programs (emulator, fp32) against the fp64 oracle, stage by stage, at scale 1 and 10."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _cases import load, as_float, rms
from test_kernel_emul import emul_subband_gsc_chain
from oracle import ds_oracle as O

g = load("g22_subbandgsc_pf_rec1_1")
x16 = g["x"]
M, FL = 4, 256
mic = O.OracleMicArray(arrayType="circular", r=0.032, M=M, n_fft=512)
g12 = load("g12_subbandgsc_rec1")
coef = np.ascontiguousarray(g12["delay_filter"], dtype=np.float32)
Fn = O.gen_noise_msc(O.OracleMicArray(arrayType="circular", r=0.032, M=M), 2 * FL)[:, 1, 2]
for scale in (1.0, 10.0):
    x = (x16.astype(np.float32) / 32768.0 * np.float32(scale)).astype(np.float32)
    out, bm, p, al = emul_subband_gsc_chain(x, M, FL, coef, Fn, False)
    o = O.OracleSubbandGSC(mic, frameLen=FL)
    ro, rfix, rbm, rp, ral = o.process(x.astype(np.float64))
    T = x.shape[1] // FL
    print("scale %g: out rms err %.3e (ref rms %.3e, rel %.2e) | bm %.3e (rel %.2e) | aligned %.3e (rel %.2e) | p max %.3e median %.1e" % (
        scale, rms(out - ro), rms(ro), rms(out - ro) / rms(ro), rms(bm - rbm), rms(bm - rbm) / rms(rbm), rms(al - ral), rms(al - ral) / rms(ral),
        np.max(np.abs(p - rp)), np.median(np.abs(p - rp))))
    # per 10-block segment
    seg = 10 * FL
    e = [rms(out[i:i + seg] - ro[i:i + seg]) / (rms(ro[i:i + seg]) + 1e-30) for i in range(0, out.shape[0], seg)]
    print("   out rel err per 10 blocks:", " ".join("%.1e" % v for v in e))
    dp = np.abs(p - rp)
    print("   p diff per 10 blocks (max):", " ".join("%.1e" % dp[:, i:i + 10].max() for i in range(0, T, 10)))
    if "output" in g.files and scale == 10.0:
        print("   oracle vs reference fixture: %.3e" % rms(ro - g["output"]))
