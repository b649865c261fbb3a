"""BASELINE config 4 on one GPU: WPE (2 taps) -> adaptive MVDR -> SPP gain, 8 mics, 1024-FFT / 512-hop, 1024 utterances per GPU
(8192 over 8 GPUs), through the DS_ALGO_WPE_MVDR chain handle with device-resident I/O.  Prints one JSON line per regime.
Algorithmic bytes per frame follow SURVEY.md section 8(d), cfg4 with N = 2 taps: bytes(T) = 18 432 + 2 * 2 197 656 / T."""
import argparse, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M, NFFT, HOP, FS = 8, 1024, 512, 16000
S_STATE = 2197656
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--hops", type=int, default=78, help="hops of audio per utterance in the timed region")
    args = ap.parse_args()
    import torch
    from distantspeech_amd import BatchEngine, _lib as L
    from distantspeech_amd.mic_array import MicArray
    dev = torch.device("cuda", 0)
    B = args.batch
    mic = MicArray(arrayType="circular", r=0.05, M=M, n_fft=NFFT)
    ang = np.array([197.0, 0.0]) / 180 * np.pi
    tao = -1 * mic.r * np.cos(ang[1]) * np.cos(ang[0] - mic.gamma) / mic.c
    a = np.exp(-1j * (2 * np.pi * np.arange(NFFT // 2 + 1) * FS / NFFT)[:, None] * tao[None, :])
    for T in (1, 39):
        K = args.hops // T
        W = max(2, 8 // T)
        Ltot = (K + W) * T * HOP
        g = torch.Generator(device=dev); g.manual_seed(4)
        x = torch.randn((B, M, Ltot), device=dev, generator=g) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(L.ALGO_WPE_MVDR, M, NFFT, HOP, batch=B, device=0, filter_len=2)
        eng.set_steering(a); eng.set_method(L.METHOD_MVDR)
        xp, yp = x.data_ptr(), y.data_ptr()
        torch.cuda.synchronize()
        run = lambda first, n: eng.process_device_seq(xp + 4 * first * T * HOP, L.LAYOUT_CHANNELS_SAMPLES, M * Ltot, Ltot, T * HOP,
                                                      T * HOP, n, yp + 4 * first * T * HOP, Ltot, T * HOP, graph=0)
        run(0, W); eng.synchronize()
        eng.timing_begin(); run(W, K); ms = eng.timing_end()
        assert bool(torch.isfinite(y[:, W * T * HOP:]).all())
        frames = B * K * T
        bytes_frame = M * HOP * 4 + HOP * 4 + 2.0 * S_STATE / T
        achieved = frames * bytes_frame / (ms * 1e-3) / 1e9
        print(json.dumps({"metric": "enhanced frames/sec (8-mic, 1024-FFT, WPE+MVDR+gain)", "value": round(frames / (ms * 1e-3), 1),
                          "unit": "frames/s", "n_gpus": 1, "steps": K, "ms_per_step": round(ms / K, 4), "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "BASELINE cfg4 chain (DS_ALGO_WPE_MVDR), batch=%d per GPU, %d hop(s) per call, 5 launches per call" % (B, T),
                                     "n_mics": M, "nfft": NFFT, "hop": HOP, "wpe_taps": 2, "wpe_delay": 4},
                          "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_frame": bytes_frame}}), flush=True)
        del x, y, eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
