"""debug: which stage of the cfg4 chain goes non-finite on a period-3 frame sequence (bench rounds of 3 steps replayed)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import distantspeech_amd as ds
from distantspeech_amd import _lib as L
from oracle import ds_oracle as O
M, nfft, hop = 8, 1024, 512
K = nfft // 2 + 1
omic = O.OracleMicArray(arrayType="circular", r=0.05, M=M, n_fft=nfft)
B = 48
x = np.stack([O.synth_utterance(100 + b, hop * 4, omic) for b in range(B)])          # [B, M, 4 hop]
tr = ds.BatchEngine(L.ALGO_TRANSFORM, M, nfft, hop, batch=B)
D = tr.stft(np.ascontiguousarray(x.transpose(0, 2, 1)), L.LAYOUT_SAMPLES_CHANNELS)   # [B, T=4, K, M]
print("D", D.shape, np.isfinite(D).all())
wpe = ds.BatchEngine(L.ALGO_WPE, M, nfft, batch=B, filter_len=2, rls_lambda=0.998)
mc = ds.BatchEngine(L.ALGO_MCMCRA, M, nfft, batch=B)
ring = [np.zeros((B, K, M), np.complex64) for _ in range(4)]
seq = [0] + [1, 2, 3] * 120
for n, t in enumerate(seq):
    d = D[:, t]
    ring.append(d); xd = ring.pop(0)
    E = wpe.wpe_update(xd[:, None], d[:, None])[:, 0]
    if not np.isfinite(E).all():
        bad = np.argwhere(~np.isfinite(E))
        print("WPE output non-finite at frame", n, "first", bad[0], "count", len(bad))
        st = wpe.op_state_raw()
        print("state finite", np.isfinite(st).all())
        break
    if n % 30 == 0:
        print(n, "E rms %.3e" % np.sqrt(np.mean(np.abs(E) ** 2)), "state absmax %.3e" % np.abs(wpe.op_state_raw()).max())
else:
    print("WPE finite through", len(seq))
