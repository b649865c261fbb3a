import numpy as np, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import distantspeech_amd as ds
g = np.load("tests/golden/g8_subband.npz")
x, d, pp = g["x"], g["d"], g["p"]
lms = ds.SubbandLMS(filter_len=2, num_bands=512, mu=0.1)
for n in range(6):
    e = lms.update(x[n], d[n], p=pp[n])[0]
    ref = g["e_lms"][n]
    err = np.abs(e - ref)
    print(n, float(np.max(err)), int(np.argmax(err)), np.nonzero(err > 1e-4)[0][:10])
