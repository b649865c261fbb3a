"""the WPE recursion's fp32 margin over a real sample instead of two seeds.
  A  cfg4 chain (WPE 8 x 2 taps -> McMcra -> MVDR + gain, 8 microphones, 1024 / 512) at BASELINE's chunk length — three 10 s chunks of 312
     hops with the state carried — over 32 utterances (seeds 40 .. 71), every 100-frame segment against the fp64 oracle, relative to the
     segment's own RMS;
  B  the wide-tap kernel (4 x 20 taps, 256 bands / hop 64) on the 32 s stationary, strongly reverberant stream over 32 seeds, 17 bins, every
     250-frame segment against the fp64 oracle core, in fp32 (default) and with DS_PARAM_WPE_FP64.
The oracle legs run in worker processes started BEFORE this process touches the GPU.  Writes one JSON line per leg to $DS_PARITY_LOG
(default gpurun_out/r06_wpe_sample.jsonl).   usage: python scratch/wpe_sample.py [n_utterances=32]"""
import json, os, sys, time
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

N_UTT = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M, NFFT, HOP, T = 8, 1024, 512, 312
C, NT, NB, WHOP, DL = 4, 20, 256, 64, 4


def rms(a):
    return float(np.sqrt(np.mean(np.abs(a) ** 2)))


def oracle_cfg4(seed):
    from oracle import ds_oracle as O
    from _cases import ANGLE, oracle_mic
    omic = oracle_mic(M, NFFT)
    x = O.synth_utterance(seed, 3 * T * HOP, omic)
    return O.OracleWpeMvdrPostfilter(omic, nfft=NFFT, hop=HOP).process(x, ANGLE).astype(np.float32)


def reverberant(seed, L, Cn, tail=3000, decay=600.0):          # tests/test_gpu_wpe_wide.py
    rng = np.random.default_rng(seed)
    n = np.arange(L + tail)
    s = rng.standard_normal(L + tail) * 0.1 * (0.25 + 0.75 * np.abs(np.sin(2 * np.pi * n / 16000 * 2.7))) * (np.sin(2 * np.pi * n / 16000 * 0.9) > -0.5)
    x = np.empty((L, Cn))
    for c in range(Cn):
        h = rng.standard_normal(tail) * np.exp(-np.arange(tail) / decay) * 0.2
        h[0] = 1.0
        x[:, c] = np.convolve(s, h)[tail:tail + L]
    return (x + 0.002 * rng.standard_normal(x.shape)).astype(np.float32)


def stream_spectra(seed):
    from oracle import ds_oracle as O
    x = reverberant(seed, 16000 * 32, C)
    x = x * (0.05 / rms(x))
    Dn = O.OracleTransform(channel=C, n_fft=NB, hop_length=WHOP).stft(x)
    ks = np.linspace(1, NB // 2 - 1, 17).astype(int)
    D = np.ascontiguousarray(Dn[ks].transpose(1, 0, 2))                          # [T, 17, C]
    Xd = np.concatenate([np.zeros((DL, 17, C), complex), D[:-DL]])
    return D.astype(np.complex64), Xd.astype(np.complex64)


def oracle_stream(seed):
    from oracle import ds_oracle as O
    D, Xd = stream_spectra(seed)
    o = O.OracleWpe(channels=C, filter_len=NT, num_bands=32, delay=DL)
    return np.stack([o.update_fd(Xd[t], D[t]) for t in range(D.shape[0])]).astype(np.complex64)


def main():
    log = os.environ.get("DS_PARITY_LOG") or os.path.join(ROOT, "gpurun_out", "r06_wpe_sample.jsonl")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as ex:           # oracle legs first: nothing here has touched the GPU yet
        fa = [ex.submit(oracle_cfg4, 40 + b) for b in range(N_UTT)]
        fb = [ex.submit(oracle_stream, 11 + s) for s in range(N_UTT)]
        refs_a = np.stack([f.result() for f in fa])
        refs_b = np.stack([f.result() for f in fb])
    print("oracle legs: %.0f s" % (time.time() - t0), flush=True)
    import distantspeech_amd as ds
    from distantspeech_amd import _lib as L
    from oracle import ds_oracle as O
    from _cases import ANGLE, oracle_mic
    # ---- A ----
    omic = oracle_mic(M, NFFT)
    x = np.stack([O.synth_utterance(40 + b, 3 * T * HOP, omic) for b in range(N_UTT)])
    obj = ds.WpeMvdrPostfilter(ds.MicArray(arrayType="circular", r=omic.r, M=M, n_fft=NFFT), frameLen=NFFT, hop=HOP, batch=N_UTT)
    y = np.concatenate([obj.process(x[:, :, c * T * HOP:(c + 1) * T * HOP], ANGLE)["data"] for c in range(3)], axis=1)
    n = 100 * HOP
    seg = np.array([[rms(y[b, i:i + n] - refs_a[b, i:i + n]) / rms(refs_a[b, i:i + n]) for i in range(0, refs_a.shape[1] - n + 1, n)] for b in range(N_UTT)])
    worst = seg.max(axis=1)
    rec = dict(test="cfg4_baseline_chunks_sample", utterances=N_UTT, segments_per_utterance=int(seg.shape[1]), worst_segment_rel_rms_per_utterance=[float(v) for v in worst],
               max=float(worst.max()), median=float(np.median(worst)), p90=float(np.percentile(worst, 90)), segments_over_1e4=int((seg > 1e-4).sum()),
               segments_total=int(seg.size), abs_rms_max=float(max(rms(y[b] - refs_a[b]) for b in range(N_UTT))), ref_rms=rms(refs_a))
    print(json.dumps(rec), flush=True)
    open(log, "a").write(json.dumps(rec) + "\n")
    del obj
    # ---- A2: the same chain with the RLS-WPE recursion in double (DS_PARAM_WPE_FP64): is the WPE stage what the margin is spent on? ----
    obj = ds.WpeMvdrPostfilter(ds.MicArray(arrayType="circular", r=omic.r, M=M, n_fft=NFFT), frameLen=NFFT, hop=HOP, batch=N_UTT)
    obj._eng.set_param_i(L.PARAM_WPE_FP64, 1)
    y = np.concatenate([obj.process(x[:, :, c * T * HOP:(c + 1) * T * HOP], ANGLE)["data"] for c in range(3)], axis=1)
    seg = np.array([[rms(y[b, i:i + n] - refs_a[b, i:i + n]) / rms(refs_a[b, i:i + n]) for i in range(0, refs_a.shape[1] - n + 1, n)] for b in range(N_UTT)])
    worst = seg.max(axis=1)
    rec = dict(test="cfg4_baseline_chunks_sample_wpe_fp64", utterances=N_UTT, worst_segment_rel_rms_per_utterance=[float(v) for v in worst],
               max=float(worst.max()), median=float(np.median(worst)), p90=float(np.percentile(worst, 90)), segments_over_1e4=int((seg > 1e-4).sum()),
               segments_total=int(seg.size), abs_rms_max=float(max(rms(y[b] - refs_a[b]) for b in range(N_UTT))))
    print(json.dumps(rec), flush=True)
    open(log, "a").write(json.dumps(rec) + "\n")
    del obj
    if os.environ.get("WPE_SAMPLE_ONLY_A"):
        return
    # ---- B ----
    Ds, Xds = zip(*[stream_spectra(11 + s) for s in range(N_UTT)])
    D, Xd = np.stack(Ds), np.stack(Xds)                                            # [B, T, 17, C]
    Tn = D.shape[1]
    for mode in ("fp32", "fp64"):
        eng = ds.BatchEngine(L.ALGO_WPE, C, 32, batch=N_UTT, filter_len=NT, rls_lambda=0.998)
        if mode == "fp64":
            eng.set_param_i(L.PARAM_WPE_FP64, 1)
        rel = []
        for a in range(0, Tn, 250):
            b = min(Tn, a + 250)
            err = eng.wpe_update(Xd[:, a:b], D[:, a:b])
            rel.append([rms(err[u] - refs_b[u, a:b]) / rms(refs_b[u, a:b]) for u in range(N_UTT)])
        rel = np.array(rel).T                                                      # [utterance, segment]
        worst = rel.max(axis=1)
        wav = rel * 0.05                                                           # the streams are scaled to 0.05 RMS: relative error x level ~ waveform RMS error
        rec = dict(test="wpe_wide_32s_stream_sample_" + mode, streams=N_UTT, frames=int(Tn), worst_segment_rel_per_stream=[float(v) for v in worst],
                   max=float(worst.max()), median=float(np.median(worst)), p90=float(np.percentile(worst, 90)), last_segment_rel_median=float(np.median(rel[:, -1])),
                   segments_over_1e4_relative=int((rel > 1e-4).sum()), segments_total=int(rel.size), worst_wav_rms_at_0p05_input=float(wav.max()))
        print(json.dumps(rec), flush=True)
        open(log, "a").write(json.dumps(rec) + "\n")
        eng.close()


if __name__ == "__main__":
    main()
