"""Experiment: the cfg5 chain as G independent handles of B / G utterances each (own streams), calls interleaved, vs one handle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from distantspeech_amd import BatchEngine, _lib as L
from distantspeech_amd.mic_array import MicArray, compute_tau
from distantspeech_amd.ops import McSpp
from distantspeech_amd.subband_gsc import fractional_delay_filter_bank
M, NFFT, HOP = 6, 512, 256
dev = torch.device("cuda", 0)
B = 2048
mic = MicArray(arrayType="circular", r=0.05, M=M, n_fft=NFFT)
tau = compute_tau(mic, np.array([197.0, 0.0]) / 180 * np.pi)
fir = fractional_delay_filter_bank(np.array(-(tau - np.max(tau)))[:, 0] * mic.fs)
for T, K in ((1, 64), (62, 2)):
    for G in (1, 2, 4):
        W = 2
        Ltot = (K + W) * T * HOP
        Bg = B // G
        xs = [torch.randn((Bg, M, Ltot), device=dev) * 0.05 for _ in range(G)]
        ys = [torch.empty((Bg, Ltot), device=dev) for _ in range(G)]
        engs = []
        for g in range(G):
            e = BatchEngine(L.ALGO_SUBBAND_GSC, M, NFFT, HOP, batch=Bg, device=0, filter_len=2, rls_lambda=0.998)
            e.chain_set_aux(L.CHAIN_AUX_FIR, fir)
            e.chain_set_aux(L.CHAIN_AUX_COHERENCE, McSpp.diffuse_coherence(M, NFFT))
            engs.append(e)
        def run(first, n):
            for i in range(first, first + n):
                for g in range(G):
                    engs[g].process_device_seq(xs[g].data_ptr() + 4 * i * T * HOP, L.LAYOUT_CHANNELS_SAMPLES, M * Ltot, Ltot, T * HOP,
                                               T * HOP, 1, ys[g].data_ptr() + 4 * i * T * HOP, Ltot, T * HOP, graph=0)
        run(0, W)
        for e in engs: e.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(W, K)
        for e in engs: e.synchronize()
        dt = time.perf_counter() - t0
        print("T=%d groups=%d: %.2f M frames/s (%.1f us per step)" % (T, G, B * K * T / dt / 1e6, dt / K * 1e6), flush=True)
        del xs, ys, engs
        torch.cuda.empty_cache()
