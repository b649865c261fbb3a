import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import distantspeech_amd as ds
rng = np.random.default_rng(31)
M, FL, T, B = 4, 256, 12, 3
x = (rng.standard_normal((B, T * FL, M)) * 0.05).astype(np.float32)
x[:, :, 1:] += 0.6 * x[:, :, :1]
mic = ds.MicArray(arrayType="circular", r=0.032, M=M, n_fft=512)
full = ds.FDGSC(mic, frameLen=FL, batch=B).process(x)
names = ["out", "p", "fix", "fix_d", "bm", "al", "al_d"]
for b in range(B):
    one = ds.FDGSC(mic, frameLen=FL).process(x[b])
    print(b, [(n, float(np.abs(a - f[b]).max())) for n, a, f in zip(names, one, full)])
