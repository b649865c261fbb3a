import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import distantspeech_amd as ds
from distantspeech_amd import _lib as L
from _cases import DeviceBuffers
M, nfft, hop, B, T, n_calls, rounds = 8, 1024, 512, 5, 2, 3, 4
Ltot = T * hop * n_calls * rounds
dv = DeviceBuffers()
xd = dv.upload((np.random.default_rng(21).standard_normal((B, M, Ltot)) * 0.05).astype(np.float32))
mic = ds.MicArray(arrayType="circular", r=0.05, M=M, n_fft=nfft)
ang = np.array([197.0, 0.0]) / 180 * np.pi
tao = -1 * mic.r * np.cos(ang[1]) * np.cos(ang[0] - mic.gamma) / mic.c
steer = np.exp(-1j * (2 * np.pi * np.arange(nfft // 2 + 1) * 16000 / nfft)[:, None] * tao[None, :])
outs = []
for n_parts, graph in ((1, 0), (2, 0), (2, 1), (1, 0), (5, 0)):
    e = ds.BatchEngine(L.ALGO_WPE_MVDR, M, nfft, hop, batch=B, device=0, filter_len=2)
    e.set_steering(steer); e.set_method(L.METHOD_MVDR); e.set_wpe_delay(3); e.set_split(n_parts)
    yd = dv.zeros(B * Ltot * 4)
    seg = T * hop * n_calls
    for r in range(rounds):
        e.process_device_seq(xd + 4 * r * seg, L.LAYOUT_CHANNELS_SAMPLES, M * Ltot, Ltot, T * hop, T * hop, n_calls, yd + 4 * r * seg, Ltot, T * hop, graph=graph)
    e.synchronize()
    y = dv.download(yd, (B, Ltot))
    outs.append(y)
    e.close()
for i, y in enumerate(outs[1:], 1):
    d = np.abs(y - outs[0])
    first = [int(np.argmax(d[b] > 0)) // hop if d[b].max() > 0 else -1 for b in range(B)]
    print("variant", i, "max diff per utterance", d.max(axis=1), "first differing hop", first)
