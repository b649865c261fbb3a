"""debug: the cfg4 chain taken apart (separate handles) on the bench's replayed hops: which stage's OUTPUT goes non-finite"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import distantspeech_amd as ds
from distantspeech_amd import _lib as L
from distantspeech_amd.mic_array import MicArray
be = bench.GpuBackend(0, 1)
w = bench.WORKLOADS["cfg4"]
M, nfft, hop, B = 8, 1024, 512, 48
K = nfft // 2 + 1
x = be.synth(w, B, 4 * hop, 0).cpu().numpy()                       # [B, M, 4 hop]: the bench's own input (seed 0)
mic = MicArray(arrayType="circular", r=w["r"], M=M, n_fft=nfft)
ang = np.array(bench.ANGLE_DEG) / 180.0 * np.pi
tao = -1 * mic.r * np.cos(ang[1]) * np.cos(ang[0] - mic.gamma) / mic.c
a = np.exp(-1j * (2 * np.pi * np.arange(K) * 16000 / nfft)[:, None] * tao[None, :])
tr = ds.BatchEngine(L.ALGO_TRANSFORM, M, nfft, hop, batch=B)
wpe = ds.BatchEngine(L.ALGO_WPE, M, nfft, batch=B, filter_len=2, rls_lambda=0.998)
mc = ds.BatchEngine(L.ALGO_MCMCRA, M, nfft, batch=B)
af = ds.BatchEngine(L.ALGO_ADAPTIVE_FRAMES, M, nfft, batch=B)
af.set_steering(a); af.set_method(L.METHOD_MVDR)
ring = [np.zeros((B, K, M), np.complex64) for _ in range(4)]
seq = [0] + [1, 2, 3] * 100
for n, t in enumerate(seq):
    d = tr.stft(np.ascontiguousarray(x[:, :, t * hop:(t + 1) * hop].transpose(0, 2, 1)), L.LAYOUT_SAMPLES_CHANNELS)[:, 0]
    ring.append(d); xd = ring.pop(0)
    E = wpe.wpe_update(xd[:, None], d[:, None])
    p, G = mc.mcmcra_estimate(E)
    Y = af.adaptive_frames(E, G)
    fin = dict(E=np.isfinite(E).all(), p=np.isfinite(p).all(), G=np.isfinite(G).all(), Y=np.isfinite(Y).all())
    if n % 30 == 0 or not all(fin.values()):
        print(n, fin, "E rms %.3e G mean %.3f Y rms %.3e" % (np.sqrt(np.mean(np.abs(E) ** 2)), G.mean(), np.sqrt(np.nanmean(np.abs(Y) ** 2))))
    if not all(fin.values()):
        for nm, arr in (("E", E), ("p", p), ("G", G), ("Y", Y)):
            bad = np.argwhere(~np.isfinite(arr))
            if len(bad):
                print(" ", nm, "bad count", len(bad), "first", bad[0], "utterances", sorted(set(bad[:, 0].tolist()))[:6], "bins", sorted(set(bad[:, 2].tolist()))[:10])
        b0, k0 = np.argwhere(~np.isfinite(Y))[0][[0, 2]]
        print("  at (b,k)=", b0, k0, "E", E[b0, 0, k0], "G", G[b0, 0, k0], "p", p[b0, 0, k0])
        st = af.op_state()[b0][:, k0]
        print("  adaptive op state at that bin", st)
        break
