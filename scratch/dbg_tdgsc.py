import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import distantspeech_amd as ds
from _cases import load, as_float, rms
from oracle import ds_oracle as O
for name in ["rec1_pf", "synth_m6_pf"]:
    g = load("g15_tdgsc_" + name)
    M, FL, pf = [int(v) for v in g["params"]]
    x = as_float(g["x"]).T
    mic = ds.MicArray(arrayType="circular", r=float(g["r"]), M=M, n_fft=512)
    tg = ds.TDGSC(mic, frameLen=FL, angle=[197, 0])
    out, p, bm = tg.process(x, postfilter=True)
    ref = g["output"]
    nb = len(out) // FL
    eb = np.array([rms(out[i*FL:(i+1)*FL] - ref[i*FL:(i+1)*FL]) for i in range(nb)])
    rb = np.array([rms(ref[i*FL:(i+1)*FL]) for i in range(nb)])
    print(name, "total rel", rms(out-ref)/rms(ref))
    print(" per-block rel err:", np.array2string(eb/np.maximum(rb,1e-9), precision=1, max_line_width=200))
