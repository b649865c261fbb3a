"""BASELINE cfg4 / cfg5 at their full per-GPU sizes: 10 s chunks, 3 chunks, whole batch — checks that nothing overflows at
tens of GB of device buffers and reports the chunked rates."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from distantspeech_amd import BatchEngine, _lib as L
from distantspeech_amd.mic_array import MicArray, compute_tau
from distantspeech_amd.ops import McSpp
from distantspeech_amd.subband_gsc import fractional_delay_filter_bank
dev = torch.device("cuda", 0)
ang = np.array([197.0, 0.0]) / 180 * np.pi

def run(name, eng, B, M, hop, T, chunks=3):
    n = T * hop
    g = torch.Generator(device=dev); g.manual_seed(1)
    y = torch.empty((B, n), device=dev)
    rates = []
    for c in range(chunks):
        x = torch.randn((B, M, n), device=dev, generator=g) * 0.05
        torch.cuda.synchronize()
        eng.synchronize(); eng.timing_begin()
        eng.process_device_seq(x.data_ptr(), L.LAYOUT_CHANNELS_SAMPLES, M * n, n, n, n, 1, y.data_ptr(), n, n, graph=0)
        ms = eng.timing_end()
        ok = bool(torch.isfinite(y).all()) and float(y.abs().max()) > 0
        rates.append(B * T / ms / 1e3)
        print("%s chunk %d: %.1f ms, %.2f M frames/s, finite+nonzero=%s, out rms %.4f" % (name, c, ms, rates[-1], ok, float(y.pow(2).mean().sqrt())), flush=True)
        del x
    print("%s: device memory in use %.1f GB" % (name, torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9), flush=True)

# cfg5: 6 mics, 512 bands, 625 blocks (10 s) per chunk, 2048 utterances
M, NFFT, HOP = 6, 512, 256
mic = MicArray(arrayType="circular", r=0.05, M=M, n_fft=NFFT)
tau = compute_tau(mic, ang)
eng = BatchEngine(L.ALGO_SUBBAND_GSC, M, NFFT, HOP, batch=2048, device=0, filter_len=2, rls_lambda=0.998)
eng.chain_set_aux(L.CHAIN_AUX_FIR, fractional_delay_filter_bank(np.array(-(tau - np.max(tau)))[:, 0] * mic.fs))
eng.chain_set_aux(L.CHAIN_AUX_COHERENCE, McSpp.diffuse_coherence(M, NFFT))
run("cfg5 B=2048 T=625", eng, 2048, M, HOP, 625)
eng.close(); del eng; torch.cuda.empty_cache()
# cfg4: 8 mics, 1024-FFT, 312 hops (10 s) per chunk, 1024 utterances
M, NFFT, HOP = 8, 1024, 512
mic = MicArray(arrayType="circular", r=0.05, M=M, n_fft=NFFT)
tao = -1 * mic.r * np.cos(ang[1]) * np.cos(ang[0] - mic.gamma) / mic.c
a = np.exp(-1j * (2 * np.pi * np.arange(NFFT // 2 + 1) * 16000 / NFFT)[:, None] * tao[None, :])
eng = BatchEngine(L.ALGO_WPE_MVDR, M, NFFT, HOP, batch=1024, device=0, filter_len=2)
eng.set_steering(a); eng.set_method(L.METHOD_MVDR)
run("cfg4 B=1024 T=312", eng, 1024, M, HOP, 312)
