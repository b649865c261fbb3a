"""BASELINE config 5 on one GPU: Subband-RLS GSC (SubbandGSC structure with SubbandRLS blocking filters), 6 mics, 512 bands,
block 256, 2048 utterances per GPU (16384 over 8 GPUs), through the DS_ALGO_SUBBAND_GSC chain handle with device-resident I/O.
Algorithmic bytes per frame follow SURVEY.md section 8(d), cfg5: bytes(T) = 7 168 + 2 * 313 440 / T.  One JSON line per regime."""
import argparse, json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M, NFFT, HOP, FS = 6, 512, 256, 16000
S_STATE = 313440
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--bm", default="rls", choices=["rls", "lms"])
    args = ap.parse_args()
    import torch
    from distantspeech_amd import BatchEngine, _lib as L
    from distantspeech_amd.mic_array import MicArray, compute_tau
    from distantspeech_amd.ops import McSpp
    from distantspeech_amd.subband_gsc import fractional_delay_filter_bank
    dev = torch.device("cuda", 0)
    B = args.batch
    mic = MicArray(arrayType="circular", r=0.05, M=M, n_fft=NFFT)
    tau = compute_tau(mic, np.array([197.0, 0.0]) / 180 * np.pi)
    fir = fractional_delay_filter_bank(np.array(-(tau - np.max(tau)))[:, 0] * mic.fs)
    for T, K in ((1, 64), (62, 2)):            # one block per call (the realtime regime), then 1 s chunks
        W = 2
        Ltot = (K + W) * T * HOP
        g = torch.Generator(device=dev); g.manual_seed(5)
        x = torch.randn((B, M, Ltot), device=dev, generator=g) * 0.05
        y = torch.empty((B, Ltot), device=dev)
        eng = BatchEngine(L.ALGO_SUBBAND_GSC, M, NFFT, HOP, batch=B, device=0, filter_len=2, rls_lambda=0.998 if args.bm == "rls" else 0.0)
        eng.chain_set_aux(L.CHAIN_AUX_FIR, fir)
        eng.chain_set_aux(L.CHAIN_AUX_COHERENCE, McSpp.diffuse_coherence(M, NFFT))
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
        print(json.dumps({"metric": "enhanced frames/sec (6-mic, 512-band subband GSC, %s blocking filters)" % args.bm.upper(),
                          "value": round(frames / (ms * 1e-3), 1), "unit": "frames/s", "n_gpus": 1, "steps": K, "ms_per_step": round(ms / K, 4),
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "BASELINE cfg5 chain (DS_ALGO_SUBBAND_GSC), batch=%d per GPU, %d block(s) per call" % (B, T),
                                     "n_mics": M, "nfft": NFFT, "hop": HOP, "taps": 2},
                          "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_frame": bytes_frame}}), flush=True)
        del x, y, eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
