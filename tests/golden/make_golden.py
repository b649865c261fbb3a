#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE (wangwei2009/DistantSpeech) in this container.

TEST INFRASTRUCTURE.  Needs /root/reference (read-only mount) — it does not exist on the GPU box;
only the .npz files this script writes travel.  Nothing from the reference is copied: the fixtures
hold inputs and the reference's numeric outputs only.

    python tests/golden/make_golden.py            # (re)writes tests/golden/*.npz

Repairs applied to broken reference entry points, each recorded in the fixture's
``meta`` string:
  R1  FixedBeamformer / adaptivebeamfomer are built with ``cls.__new__`` + ``beamformer.__init__``
      + a replay of the remaining ctor body (their own ctors pass c=/fs= that
      beamformer.__init__ no longer accepts: fixedbeamformer.py:99, adaptivebeamformer.py:13).
  R2  MicArray(..., n_fft=nfft) so compute_steering_vector_from_doa fills all bins (beamformer.py:286).
  R3  adaptivebeamfomer.process / GSC.process are called ONE HOP PER CALL and concatenated
      (they hand a 2-D [K, T] array to Transform.istft, which reads 2-D as [K, channels]).
  R4  SD weights at 512-FFT via the base class beamformer.compute_weights(weightType='SD')
      (FixedBeamformer.compute_weights calls gen_noise_msc with nfft=256: shape error).
  R5  stray prints silenced (fixedbeamformer.py:68-74).
  R6  Wpe (g10, g18, g21): `awpe.Subband := Transform` — the reference builds Wpe on its Nyquist filterbank whose
      design does not terminate at the sizes of interest; the STFT grid is used instead.
  R8  McSpp with M != 4 (g11 synth_m6 only): `mcspp.mccdr = McCDR(nfft, channels=M)` — McSpp builds its McCDR with the
      default 4 channels (mcspp.py:54) and raises IndexError for other array sizes.
  R9  SubbandGSC (g12 only): `DistantSpeech.beamformer.FDGSC.DelayObj = object` before importing SubbandGSC
      (SubbandGSC.py:23 imports a DelayObj that FDGSC.py does not define).
  R7  Wpe (g10, g18, g21): `Wpe.check_input_data(xd, x)` is undefined at HEAD (awpe.py:150); defined here as the
      analogue of SubbandAF.update_input_data (SubbandAF.py:53-60): analyse both signals, set return_td = True.
  R10 compute_pmwf_weight (g19 only): the free function's `channels = Rxx.shape[0]` (beamformer.py:124) makes `u` [bins, bins, 1], which
      only multiplies with [bins, M, M] matrices where M == bins; the fixture replays its body with channels = Rxx.shape[1].
Third-party versions at generation time are recorded in every fixture.
"""
import contextlib
import glob
import io
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim  # noqa: E402

_ref_shim.install()

from scipy.io import wavfile  # noqa: E402
from scipy.signal import windows  # noqa: E402

from DistantSpeech.adaptivefilter.SubbandLMS import SubbandLMS  # noqa: E402
from DistantSpeech.adaptivefilter.SubbandLmsMc import SubbandLmsMc  # noqa: E402
from DistantSpeech.adaptivefilter.SubbandRLS import SubbandRLS  # noqa: E402
from DistantSpeech.beamformer.adaptivebeamformer import adaptivebeamfomer  # noqa: E402
from DistantSpeech.beamformer.beamformer import beamformer  # noqa: E402
from DistantSpeech.beamformer.fixedbeamformer import FixedBeamformer  # noqa: E402
from DistantSpeech.beamformer.GSC import GSC  # noqa: E402
from DistantSpeech.beamformer.MicArray import MicArray  # noqa: E402
from DistantSpeech.noise_estimation.mc_mcra import McMcra  # noqa: E402
from DistantSpeech.noise_estimation.mcra import NoiseEstimationMCRA  # noqa: E402
from DistantSpeech.noise_estimation.mcspp_base import McSppBase  # noqa: E402
from DistantSpeech.noise_estimation.omlsa_multi import NsOmlsaMulti  # noqa: E402
from DistantSpeech.transform.transform import Transform  # noqa: E402

VERSIONS = "numpy=%s scipy=%s python=%s" % (np.__version__, scipy.__version__, sys.version.split()[0])
ANGLE = np.array([197, 0]) / 180 * np.pi


def save(name, meta, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(meta + " | " + VERSIONS), **arrays)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def rec1_int16(start_s=3.0, dur_s=3.0):
    """[4, L] int16 excerpt of the reference's own test recording (example/test_audio/rec1)."""
    files = sorted(glob.glob(os.path.join(_ref_shim.REFERENCE_ROOT, "example/test_audio/rec1/*.wav")))
    chans = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for f in files:
            sr, d = wavfile.read(f)
            assert sr == 16000 and d.dtype == np.int16
            chans.append(d[int(start_s * sr): int((start_s + dur_s) * sr)])
    x = np.stack(chans)
    L = (x.shape[1] // 256) * 256
    return x[:, :L]


def synth(seed, M, L):
    rng = np.random.default_rng(seed)
    t = np.arange(L) / 16000.0
    s = np.sin(2 * np.pi * 440 * t) * (np.sin(2 * np.pi * 1.5 * t) > 0) * 0.2
    x = rng.standard_normal((M, L)) * 0.05 + s[None, :] * (1 + 0.1 * np.arange(M))[:, None]
    return x.astype(np.float32)


def make_adaptive(mic, nfft, hop):
    """R1: adaptivebeamfomer without its broken ctor (adaptivebeamformer.py:13-42 replayed)."""
    obj = adaptivebeamfomer.__new__(adaptivebeamfomer)
    beamformer.__init__(obj, mic, frame_len=nfft, hop=hop, nfft=nfft)
    obj.M = mic.M
    obj.gamma = mic.gamma
    obj.window = windows.hann(obj.frameLen, sym=False)
    obj.win_scale = np.sqrt(1.0 / obj.window.sum() ** 2)
    obj.freq_bin = np.linspace(0, obj.half_bin - 1, obj.half_bin)
    obj.omega = 2 * np.pi * obj.freq_bin * obj.fs / obj.nfft
    obj.H = np.ones([obj.M, obj.half_bin], dtype=complex) / obj.M
    obj.angle = np.array([0, 0]) / 180 * np.pi
    obj.method = "MVDR"
    obj.frameCount = 0
    obj.calc = 0
    obj.estPos = None
    obj.Rvv = np.zeros((obj.half_bin, obj.M, obj.M), dtype=complex)
    obj.Rvv_inv = np.zeros((obj.half_bin, obj.M, obj.M), dtype=complex)
    obj.Ryy = np.zeros((obj.half_bin, obj.M, obj.M), dtype=complex)
    obj.AlgorithmList = ["src", "DS", "MVDR", "TFGSC"]
    obj.AlgorithmIndex = 0
    obj.transformer = Transform(n_fft=obj.nfft, hop_length=obj.hop, channel=obj.M)
    obj.mcra = NoiseEstimationMCRA(nfft=obj.nfft)
    obj.update_noise_psd_flag = 0
    return obj


def make_fixed(mic, nfft, hop, angle_deg, weightType):
    """R1 + R4: FixedBeamformer without its broken ctor; weights from the base class."""
    obj = FixedBeamformer.__new__(FixedBeamformer)
    beamformer.__init__(obj, mic, frame_len=nfft, hop=hop, nfft=nfft)
    obj.angle = list(angle_deg)
    obj.AlgorithmList = ["src", "DS", "MVDR"]
    obj.AlgorithmIndex = 0
    obj.W = beamformer.compute_weights(obj, look_angle=list(angle_deg), weightType=weightType)
    return obj


# ------------------------------------------------------------------------------------------------
def g1_transform():
    rng = np.random.default_rng(11)
    for (nfft, hop, M) in [(512, 256, 4), (1024, 512, 2), (256, 128, 1)]:
        L = hop * 12
        x = (rng.standard_normal((L, M)) * 0.1).astype(np.float32)
        t1 = Transform(n_fft=nfft, hop_length=hop, channel=M)
        Y_one = t1.stft(x.astype(np.float64))
        y_one = t1.istft(Y_one)
        # chunked: 1 hop, then 3 hops, then the rest
        t2 = Transform(n_fft=nfft, hop_length=hop, channel=M)
        cuts = [0, hop, 4 * hop, L]
        Ys, ys = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            Yc = t2.stft(x[a:b].astype(np.float64))
            Ys.append(Yc)
            ys.append(np.atleast_2d(t2.istft(Yc).T).T.reshape(b - a, -1))
        Y_chunk = np.concatenate(Ys, axis=1)
        y_chunk = np.concatenate(ys, axis=0)
        assert np.array_equal(Y_one, Y_chunk)
        save("g1_transform_%d_%d_%d" % (nfft, hop, M),
             "Transform.stft/istft transform.py:430-481; chunked==one-shot verified bit-for-bit at generation",
             x=x, Y=Y_one.astype(np.complex64), y=np.asarray(y_one).reshape(L, -1),
             y_chunk=y_chunk, params=np.array([nfft, hop, M]))


def g1c_transform_quarter_hop():
    """Transform(n_fft, hop_length = n_fft / 4) (transform.py:407-428 takes any hop; the overlap carried between calls is then three
    hops): same checks as g1, for 1, 2, 4 and 5 channels."""
    rng = np.random.default_rng(13)
    for (nfft, hop, M) in [(512, 128, 2), (256, 64, 4), (1024, 256, 1), (512, 128, 5)]:
        L = hop * 21
        x = (rng.standard_normal((L, M)) * 0.1).astype(np.float32)
        t1 = Transform(n_fft=nfft, hop_length=hop, channel=M)
        Y_one = t1.stft(x.astype(np.float64))
        y_one = t1.istft(Y_one)
        t2 = Transform(n_fft=nfft, hop_length=hop, channel=M)
        cuts = [0, hop, 3 * hop, 10 * hop, L]                      # 1 hop (shorter than the overlap), 2, 7 and 11 hops
        Ys, ys = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            Yc = t2.stft(x[a:b].astype(np.float64))
            Ys.append(Yc)
            ys.append(np.atleast_2d(t2.istft(Yc).T).T.reshape(b - a, -1))
        assert np.array_equal(Y_one, np.concatenate(Ys, axis=1))
        save("g1c_transform_%d_%d_%d" % (nfft, hop, M),
             "Transform.stft/istft with hop = n_fft/4, transform.py:407-481; chunked stft == one-shot verified bit-for-bit at generation",
             x=x, Y=Y_one.astype(np.complex64), y=np.asarray(y_one).reshape(L, -1), y_chunk=np.concatenate(ys, axis=0),
             params=np.array([nfft, hop, M]))


def g1b_transform_window():
    """Transform(window=...) (transform.py:415-416): a caller-supplied window of n_fft samples (here a Hamming window, which is not
    power-complementary, so the round trip is not the identity) through analysis and synthesis."""
    rng = np.random.default_rng(12)
    nfft, hop, M = 512, 256, 2
    L = hop * 10
    x = (rng.standard_normal((L, M)) * 0.1).astype(np.float32)
    win = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(nfft) / nfft)
    t = Transform(n_fft=nfft, hop_length=hop, channel=M, window=win)
    Y = t.stft(x.astype(np.float64))
    y = t.istft(Y)
    save("g1b_transform_window", "Transform(window=hamming).stft/istft transform.py:415-416,430-481",
         x=x, window=win, Y=Y.astype(np.complex64), y=np.asarray(y).reshape(L, -1), params=np.array([nfft, hop, M]))


def g2_weights():
    for (atype, M, r, nfft, ang) in [("circular", 4, 0.032, 512, (197, 0)), ("linear", 6, 0.05, 512, (60, 0)),
                                      ("circular", 8, 0.05, 1024, (197, 10))]:
        mic = MicArray(arrayType=atype, r=r, M=M, n_fft=nfft)      # R2
        bf = beamformer(mic, frame_len=nfft, hop=nfft // 2, nfft=nfft)
        a0 = bf.compute_steering_vector_from_doa(look_angle=ang)
        Wds = bf.compute_weights(look_angle=list(ang), weightType="DS")
        Wsd = bf.compute_weights(look_angle=list(ang), weightType="SD")   # R4
        save("g2_weights_%s_M%d_%d" % (atype, M, nfft),
             "beamformer.compute_steering_vector_from_doa/compute_weights beamformer.py:267-289,338-373; R2 R4",
             a0=a0, Wds=Wds, Wsd=Wsd, Fvv=bf.Fvv, mic_loc=mic.mic_loc, gamma=mic.gamma,
             params=np.array([M, nfft, ang[0], ang[1]], dtype=np.float64), r=np.array(r))


def g3_mcra(x16):
    x = x16.astype(np.float32) / 32768.0
    tr = Transform(n_fft=512, hop_length=256, channel=1)
    D = tr.stft(x[0].astype(np.float64))[:, :, 0]
    P = np.abs(D * np.conj(D))
    for L in (15, 10):
        est = NoiseEstimationMCRA(nfft=512)
        est.L = L
        T = P.shape[1]
        lam = np.zeros((T, 257)); p = np.zeros((T, 257)); S = np.zeros((T, 257)); Smin = np.zeros((T, 257))
        for n in range(T):
            est.estimation(P[:, n].copy())
            lam[n], p[n], S[n], Smin[n] = est.lambda_d, est.p, est.S, est.Smin
        save("g3_mcra_L%d" % L, "NoiseEstimationMCRA.estimation mcra.py:27-77 on |STFT(rec1 ch0)|^2, L=%d" % L,
             P=P.T.copy(), lambda_d=lam, p=p, S=S[::8], Smin=Smin[::8])


def g4_adaptive(x16):
    cases = [("rec1", x16.astype(np.float32) / 32768.0, 4, 512, 256, 2),
             ("synth", synth(5, 4, 256 * 90), 4, 512, 256, 2),
             ("synth_ds", synth(6, 4, 256 * 40), 4, 512, 256, 1),
             ("synth_src", synth(6, 4, 256 * 40), 4, 512, 256, 0),
             ("synth_tfgsc", synth(8, 4, 256 * 60), 4, 512, 256, 3),
             ("synth_m6", synth(7, 6, 256 * 60), 6, 512, 256, 2),
             ("synth_m8_1024", synth(9, 8, 512 * 40), 8, 1024, 512, 2)]
    for name, x, M, nfft, hop, method in cases:
        mic = MicArray(arrayType="circular", r=0.032 if M == 4 else 0.05, M=M, n_fft=nfft)   # R2
        ab = make_adaptive(mic, nfft, hop)                                                 # R1
        T = x.shape[1] // hop
        ys = []
        snap = {}
        with contextlib.redirect_stdout(io.StringIO()):
            for t in range(T):                                                             # R3
                out = ab.process(x[:, t * hop:(t + 1) * hop].astype(np.float64), ANGLE, method=method)
                ys.append(np.atleast_1d(out["data"]))
                if name in ("rec1", "synth") and t in (0, 30, 61):
                    snap["Rvv_t%d" % t] = ab.Rvv.copy()
                    snap["H_t%d" % t] = ab.H.copy()
                    snap["p_t%d" % t] = ab.mcra.p.copy()
        y = np.concatenate(ys)
        save("g4_adaptive_%s" % name,
             "adaptivebeamfomer.process(method=%d) adaptivebeamformer.py:44-128 hop-by-hop; R1 R2 R3; angle=197deg"
             % method,
             x=(x16 if name == "rec1" else x), y=y, Rvv=ab.Rvv, Rvv_inv=ab.Rvv_inv, Ryy=ab.Ryy, H=ab.H,
             mcra_p=ab.mcra.p, mcra_lambda_d=ab.mcra.lambda_d,
             params=np.array([M, nfft, hop, method]), r=np.array(mic.r), **snap)


def g2b_fixed(x16):
    x = x16.astype(np.float32) / 32768.0
    mic = MicArray(arrayType="circular", r=0.032, M=4, n_fft=512)                          # R2
    for wt in ("DS", "SD"):
        fb = make_fixed(mic, 512, 256, (197, 0), wt)                                       # R1 R4
        # mirror FixedBeamformer.process body (:190-207) without the per-call weight recompute
        xt = x.T.astype(np.float64)[: 256 * 100]
        D = fb.transform.stft(xt)
        Yf = np.zeros((D.shape[0], D.shape[1], 1), dtype=complex)
        for n in range(D.shape[1]):
            Yf[:, n, 0] = fb.process_freframe(D[:, n, :])
        y = fb.transform.istft(Yf)
        save("g2b_fixed_%s" % wt,
             "FixedBeamformer.process_freframe loop + Transform fixedbeamformer.py:147-207; R1 R2 R4; angle=197deg",
             x=x16[:, : 256 * 100], y=y, W=fb.W)


def g5_mcmcra(x16):
    x = x16.astype(np.float32) / 32768.0
    for name, xx, M in (("rec1", x, 4), ("synth_m6", synth(21, 6, 256 * 60), 6)):
        tr = Transform(n_fft=512, hop_length=256, channel=M)
        D = tr.stft(xx.T.astype(np.float64))
        est = McMcra(nfft=512, channels=M)
        T = D.shape[1]
        p = np.zeros((T, 257)); G = np.zeros((T, 257)); xi = np.zeros((T, 257)); gam = np.zeros((T, 257))
        for n in range(T):
            est.estimation(D[:, n, :])
            p[n], G[n], xi[n], gam[n] = est.p, est.G, est.xi, est.gamma
        save("g5_mcmcra_%s" % name, "McMcra.estimation mc_mcra.py:179-224 frame by frame on Transform.stft output",
             x=(x16 if name == "rec1" else xx), p=p, G=G, xi=xi[::8], gamma=gam[::8],
             Phi_vv=np.moveaxis(est.Phi_vv, 2, 0), Phi_yy=np.moveaxis(est.Phi_yy, 2, 0), params=np.array([M, 512, 256]))


def g6_gsc(x16):
    x = x16.astype(np.float32) / 32768.0
    for name, xx, M, method in (("rec1", x, 4, 2), ("synth_m6", synth(31, 6, 256 * 50) * np.float32(0.1), 6, 2),
                                ("synth_m4", synth(33, 4, 256 * 50) * np.float32(0.1), 4, 2),
                                ("synth_m0", synth(32, 4, 256 * 20), 4, 0)):
        mic = MicArray(arrayType="circular", r=0.032 if M == 4 else 0.05, M=M, n_fft=512)  # R2
        with contextlib.redirect_stdout(io.StringIO()):                                    # R5
            g = GSC(mic, frameLen=512, angle=[197, 0])
        T = xx.shape[1] // 256
        ys = []
        with contextlib.redirect_stdout(io.StringIO()):
            for t in range(T):                                                             # R3
                out = g.process(xx[:, t * 256:(t + 1) * 256].astype(np.float64), ANGLE, method=method)
                ys.append(np.atleast_1d(out["data"]))
        y = np.concatenate(ys)
        save("g6_gsc_%s" % name,
             "GSC.process(method=%d) GSC.py:174-294 hop-by-hop; R2 R3 R5; angle=197deg" % method,
             x=(x16 if name == "rec1" else xx), y=y, G=g.G, spp_G=g.spp.G, spp_p=g.spp.p,
             omlsa_G=g.omlsa_multi.G, omlsa_p=g.omlsa_multi.p, omlsa_lambda_d=np.asarray(g.omlsa_multi.lambda_d),
             mcra_p=g.mcra.p, params=np.array([M, 512, 256, method]), r=np.array(mic.r))


def g7_omlsa():
    rng = np.random.default_rng(41)
    K, M, T = 257, 4, 120
    env = 1.0 + 4.0 * (np.sin(np.arange(T) / 9.0) > 0.6)
    y = (rng.chisquare(2, size=(T, K)) * 0.01 * env[:, None])
    u = (rng.chisquare(2, size=(T, K, M - 1)) * 0.01)
    est = NsOmlsaMulti(nfft=512, M=M, cal_weights=True)
    G = np.zeros((T, K)); p = np.zeros((T, K)); lam = np.zeros((T, K)); xi = np.zeros((T, K)); q = np.zeros((T, K))
    for n in range(T):
        est.estimation(y[n].copy(), u[n].copy())
        G[n], p[n], lam[n], xi[n], q[n] = est.G, est.p, est.lambda_d, est.xi_hat, est.q_hat
    save("g7_omlsa", "NsOmlsaMulti.estimation omlsa_multi.py:73-156 (cal_weights=True) on chi2 powers",
         y=y, u=u, G=G, p=p, lambda_d=lam, xi_hat=xi[::8], q_hat=q[::8])


def g8_subband():
    rng = np.random.default_rng(51)
    K, T, M = 257, 60, 3
    x = (rng.standard_normal((T, K)) + 1j * rng.standard_normal((T, K))) * 0.3
    h0, h1 = 0.7 - 0.2j, -0.3 + 0.1j
    d = np.conj(h0) * x + np.conj(h1) * np.vstack([np.zeros((1, K)), x[:-1]]) + 0.01 * rng.standard_normal((T, K))
    pp = rng.uniform(0.0, 1.0, size=(T, K))
    lms = SubbandLMS(filter_len=2, num_bands=512, mu=0.1)
    rls = SubbandRLS(filter_len=2, num_bands=512)
    e_l = np.zeros((T, K), dtype=complex); e_r = np.zeros((T, K), dtype=complex)
    for n in range(T):
        e_l[n], _ = lms.update(x[n].copy(), d[n].copy(), p=pp[n].copy())
        e_r[n], _ = rls.update(x[n].copy(), d[n].copy())
    xm = (rng.standard_normal((T, K, M)) + 1j * rng.standard_normal((T, K, M))) * 0.3
    dm = np.sum(xm * np.array([0.5, -0.25j, 0.1])[None, None, :], axis=2)
    mc = SubbandLmsMc(filter_len=2, num_bands=512, channel=M, mu=0.1)
    e_m = np.zeros((T, K), dtype=complex)
    for n in range(T):
        e_m[n], _ = mc.update(xm[n][:, None, :].copy(), dm[n].copy(), p=pp[n][:, None].copy())  # x_n is [K, 1, C] (SubbandLmsMc.py:93)
    save("g8_subband", "SubbandLMS/SubbandRLS/SubbandLmsMc.update with complex [K] inputs "
         "(SubbandLMS.py:28-84, SubbandRLS.py:44-71, SubbandLmsMc.py:144-191)",
         x=x, d=d, p=pp, e_lms=e_l, W_lms=lms.W, P_lms=lms.P, e_rls=e_r, W_rls=rls.W, P_rls=rls.P,
         xm=xm, dm=dm, e_mc=e_m, W_mc=mc.W, P_mc=mc.P)


def g9_mcsppbase(x16):
    x = x16.astype(np.float32) / 32768.0
    for name, xx, M in (("rec1", x, 4), ("synth_m6", synth(61, 6, 256 * 60), 6)):
        tr = Transform(n_fft=512, hop_length=256, channel=M)
        D = tr.stft(xx.T.astype(np.float64))
        est = McSppBase(nfft=512, channels=M)
        T = D.shape[1]
        p = np.zeros((T, 257)); xi = np.zeros((T, 257)); w = np.zeros((T, 257, M), dtype=complex)
        for n in range(T):
            est.estimation(D[:, n, :])
            p[n], xi[n], w[n] = est.p, est.xi, est.w
        save("g9_mcsppbase_%s" % name, "McSppBase.estimation + compute_pmwf_weight mcspp_base.py:220-324 frame by frame",
             x=(x16 if name == "rec1" else xx), p=p, xi=xi[::8], w=w[::4].astype(np.complex64), w_last=est.w, gamma=est.gamma,
             Phi_vv=est.Phi_vv, Phi_yy=est.Phi_yy, mcra_p=est.mcra.p, params=np.array([M, 512, 256]))


def g10_wpe():
    from DistantSpeech.dereverberation import awpe
    awpe.Subband = Transform                                                              # R6

    def check_input_data(self, xd, x):                                                    # R7
        self.return_td = True
        return np.squeeze(self.transform_x.analysis(xd)), np.squeeze(self.transform_d.analysis(x))

    awpe.Wpe.check_input_data = check_input_data
    rng = np.random.default_rng(71)
    for name, C, N, D, nb in (("c4n2", 4, 2, 2, 256), ("c2n3", 2, 3, 1, 512)):
        hop = nb // 2
        T = 50
        s0 = rng.standard_normal(hop * T + 4000) * 0.1
        h = rng.standard_normal((C, 3000)) * np.exp(-np.arange(3000) / 600.0)[None, :] * 0.2
        h[:, 0] = 1.0
        x = np.stack([np.convolve(s0, h[c])[: hop * T] for c in range(C)], axis=1)
        x = (x + 0.01 * rng.standard_normal(x.shape)).astype(np.float32)
        with contextlib.redirect_stdout(io.StringIO()):
            wpe = awpe.Wpe(filter_len=N, delay=D, channels=C, num_bands=nb, hop_length=hop)
        outs = []
        for n in range(T):
            out, _ = wpe.update(x[n * hop:(n + 1) * hop].astype(np.float64))
            outs.append(np.atleast_1d(out))
        save("g10_wpe_%s" % name,
             "Wpe.update awpe.py:129-192 hop-by-hop on the STFT grid; R6 R7 (PARITY UNPINNED by the reference as shipped)",
             x=x, y=np.concatenate(outs), W=wpe.W, P=wpe.P, var=wpe.var, params=np.array([C, N, D, nb, hop]))


def g11_mcspp(x16):
    """McSpp + the notebook's online MVDR flow (example/mvdr.ipynb cell 4): estimation -> steering -> compute_mvdr_weight."""
    from DistantSpeech.noise_estimation.mcspp import McSpp
    from DistantSpeech.noise_estimation.mccdr import McCDR
    from DistantSpeech.beamformer.beamformer import steering, compute_mvdr_weight
    x = x16.astype(np.float32) / 32768.0
    for name, xx, M in (("rec1", x, 4), ("synth_m6", synth(81, 6, 256 * 70), 6), ("rec1_repeat", x[:, : 256 * 100], 4)):
        repeat = name.endswith("_repeat")                                                  # estimation(repeat=True), mcspp.py:280-282
        tr = Transform(n_fft=512, hop_length=256, channel=M)
        D = tr.stft(xx.T.astype(np.float64))
        with contextlib.redirect_stdout(io.StringIO()):
            est = McSpp(nfft=512, channels=M)
            if M != 4:
                est.mccdr = McCDR(512, channels=M)                                         # R8 (SURVEY 8c repair 7)
        T = D.shape[1]
        p = np.zeros((T, 257)); q = np.zeros((T, 257)); Yout = np.zeros((T, 257), dtype=complex)
        wp = np.zeros((T, 257, M), dtype=complex)
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            for n in range(T):
                est.estimation(D[:, n, :], repeat=repeat)
                p[n], q[n], wp[n] = est.p, est.q, est.w
                sv = steering(est.Phi_xx)
                w = compute_mvdr_weight(sv, est.Phi_vv_inv)
                Yout[n] = np.einsum('ij,ij->i', w.conj(), D[:, n, :])
        y = tr.istft(Yout.T[:, :, None])
        save("g11_mcspp_%s" % name,
             "McSpp.estimation mcspp.py:244-305 + steering/compute_mvdr_weight beamformer.py:10-31,133-155 (mvdr.ipynb cell 4)"
             + ("; R8 mccdr=McCDR(nfft, channels=M)" if M != 4 else "") + ("; repeat=True" if repeat else ""),
             x=(x16 if name == "rec1" else x16[:, : 256 * 100] if repeat else xx), p=p, q=q[::4], Yout=Yout.astype(np.complex64), y=y, w_pmwf=wp[::8].astype(np.complex64),
             steer_last=sv, w_last=w, Phi_xx=est.Phi_xx, Phi_vv_inv=est.Phi_vv_inv, Phi_vv=est.Phi_vv, params=np.array([M, 512, 256]))


def g11b_mcspp_an101():
    """ATTEMPT (round 6): the notebook's online MVDR flow (example/mvdr.ipynb cell 4) on the reference's own 8-channel recording
    (example/test_audio/an101-mtms-arrA) with McSpp's McCDR rebuilt for 8 channels (R8).  The reference does not get through it: from frame 5 on
    estimation_core inverts Phi_yy WITHOUT loading wherever xi < 0 (mcspp.py:222-228), and after six frames the recursive average of eight-channel
    outer products has rank <= 6 — numpy raises LinAlgError("Singular matrix") (on synthetic noise the same inverse is finite and meaningless,
    1e16).  No fixture can be made, so McSpp / DS_ALGO_MCSPP_MVDR stay at 2 / 4 / 6 channels; this function records the finding."""
    from DistantSpeech.noise_estimation.mcspp import McSpp
    from DistantSpeech.noise_estimation.mccdr import McCDR
    x16 = an101_int16()[:, : 256 * 92]
    M = 8
    tr = Transform(n_fft=512, hop_length=256, channel=M)
    D = tr.stft((x16.astype(np.float32) / 32768.0).T.astype(np.float64))
    with contextlib.redirect_stdout(io.StringIO()):
        est = McSpp(nfft=512, channels=M)
        est.mccdr = McCDR(512, channels=M)                                                 # R8 (SURVEY 8c repair 7)
    try:
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            for n in range(D.shape[1]):
                est.estimation(D[:, n, :])
    except np.linalg.LinAlgError as e:
        print("g11b: the reference's McSpp(channels=8) fails on an101 at frame %d: %s (mcspp.py:226) — no fixture" % (n, e))
        return
    print("g11b: the reference got through an101 with 8 channels (unexpected: numpy / scipy versions differ from the ones this was written with)")


def g12_subbandgsc(x16):
    import DistantSpeech.beamformer.FDGSC as FD
    FD.DelayObj = object                                                                   # R9 (SURVEY 8c repair 6)
    from DistantSpeech.beamformer.SubbandGSC import SubbandGSC
    from DistantSpeech.noise_estimation.mccdr import McCDR
    from DistantSpeech.adaptivefilter.SubbandRLS import SubbandRLS
    x = x16.astype(np.float32) / 32768.0
    for name, xx, M, rls in (("rec1", x[:, : 256 * 120], 4, False), ("synth_m6", synth(91, 6, 256 * 60), 6, False),
                             ("synth_m6_rls", synth(92, 6, 256 * 60), 6, True)):
        mic = MicArray(arrayType="circular", r=0.032 if M == 4 else 0.05, M=M, n_fft=512)
        with contextlib.redirect_stdout(io.StringIO()):
            g = SubbandGSC(mic, frameLen=256, angle=[197, 0])
            if M != 4:
                g.spp.mccdr = McCDR(512, channels=M)                                       # R8
            if rls:
                g.bm = [SubbandRLS(filter_len=2, num_bands=512) for _ in range(M)]          # config-5 composition (ours)
            with np.errstate(all="ignore"):
                out, fix, bm, p, al = g.process(xx.astype(np.float64).copy())
        save("g12_subbandgsc_%s" % name,
             "SubbandGSC.process(postfilter=False) SubbandGSC.py:170-262; R9 (FDGSC.DelayObj patched)"
             + ("; R8" if M != 4 else "") + ("; bm := SubbandRLS(filter_len=2) (config-5 composition defined by us)" if rls else ""),
             x=(x16[:, : 256 * 120] if name == "rec1" else xx), output=out, fix_output=fix, bm_output=bm.astype(np.float32), p=p,
             aligned_output=al.astype(np.float32), delay_filter=g.time_alignment.delay_filter, params=np.array([M, 256, int(rls)]),
             r=np.array(mic.r))


def g22_subbandgsc_postfilter(x16):
    """SubbandGSC.process(postfilter=True) (SubbandGSC.py:236-249): the five results are those of postfilter=False; the branch's only trace is
    the object's omlsa_multi.  Driven one block per call (the realtime contract, where transform_bm sees a one-block array and behaves as
    a streaming analysis) and in calls of several blocks (where the branch re-analyses the whole bm_output array in every block)."""
    import DistantSpeech.beamformer.FDGSC as FD
    FD.DelayObj = object                                                                   # R9
    from DistantSpeech.beamformer.SubbandGSC import SubbandGSC
    # the recording at ten times its level (at its own level the omlsa_multi powers, 1e-20, sit under the estimator's 1e-6 regularisers) and,
    # for the five returned signals, at its own level as well (rec1_1_lvl1: round 5)
    for name, SCALE, M, blocks_per_call in (("rec1_1", 10.0, 4, 1), ("rec1_5", 10.0, 4, 5), ("rec1_1_lvl1", 1.0, 4, 1)):
        xx = (x16.astype(np.float32) / 32768.0 * np.float32(SCALE))[:, : 256 * 60]
        mic = MicArray(arrayType="circular", r=0.032, M=M, n_fft=512)
        with contextlib.redirect_stdout(io.StringIO()):
            g = SubbandGSC(mic, frameLen=256, angle=[197, 0])
            outs = []
            with np.errstate(all="ignore"):
                for a in range(0, xx.shape[1], 256 * blocks_per_call):
                    outs.append(g.process(xx[:, a:a + 256 * blocks_per_call].astype(np.float64).copy(), postfilter=True)[0])
        om = g.omlsa_multi
        save("g22_subbandgsc_pf_%s" % name,
             "SubbandGSC.process(postfilter=True) SubbandGSC.py:170-262, %d block(s) per call, x = int16 / 32768 * scale; R9; the object's omlsa_multi after the last call" % blocks_per_call,
             x=x16[:, : 256 * 60], output=np.concatenate(outs), omlsa_G=om.G, omlsa_p=om.p, omlsa_lambda_d=np.asarray(om.lambda_d),
             omlsa_xi_hat=om.xi_hat, omlsa_q_hat=om.q_hat, params=np.array([M, 256, blocks_per_call]), r=np.array(mic.r), scale=np.array(SCALE))


def g13_tdfilters():
    from DistantSpeech.adaptivefilter.BaseFilter import BaseFilter
    from DistantSpeech.adaptivefilter.RLS import Rls
    rng = np.random.default_rng(101)
    n = 3000
    x = rng.standard_normal(n) * 0.3
    h = rng.standard_normal(40) * np.exp(-np.arange(40) / 8.0)
    d = np.convolve(x, h)[:n] + 0.01 * rng.standard_normal(n)
    nl = BaseFilter(filter_len=64, mu=0.1)
    e1 = np.zeros(n)
    for i in range(n):
        e, _ = nl.update(x[i], d[i])
        e1[i] = np.squeeze(e)
    nl2 = BaseFilter(filter_len=300, mu=0.2, normalization=False)
    e3 = np.zeros(1000)
    for i in range(1000):
        e, _ = nl2.update(x[i] * 0.1, d[i] * 0.1, p=0.5)
        e3[i] = np.squeeze(e)
    rl = Rls(filter_len=32)
    e2 = np.zeros(n)
    for i in range(n):
        e, _ = rl.update(x[i], d[i])
        e2[i] = np.squeeze(e)
    save("g13_tdfilters", "BaseFilter.update (BaseFilter.py:52-85; filter_len 64 normalised, 300 plain LMS p=0.5) and Rls.update "
         "(RLS.py:26-42; filter_len 32) sample by sample",
         x=x, d=d, e_nlms=e1, w_nlms=nl.w[:, 0], e_lms=e3, w_lms=nl2.w[:, 0], e_rls=e2, w_rls=rl.w[:, 0], P_rls=rl.P)


def g14_fdaf():
    from DistantSpeech.adaptivefilter.FastFreqLms import FastFreqLms
    from DistantSpeech.beamformer.gsc_bm import AdaptiveBlockingMatrixFilter
    from DistantSpeech.beamformer.gsc_aic import AdaptiveInterferenceCancellation
    rng = np.random.default_rng(141)

    def run(f, x, d, p, trunc):
        hop = f.hop_len
        nb = x.shape[0] // hop
        e = np.zeros(nb * hop)
        for n in range(nb):
            pn = p[n] if np.ndim(p) == 1 else p[n][:, None]
            en, w = f.update(x[n * hop:(n + 1) * hop], d[n * hop:(n + 1) * hop], p=pn, fir_truncate=trunc)
            e[n * hop:(n + 1) * hop] = en[:, 0]
        return e, np.array(w), np.array(f.W), np.array(f.P[:, 0])

    # (a) single channel, causal, unit p — plain system identification
    L, nb = 64, 40
    x = rng.standard_normal(L * nb) * 0.3
    h = rng.standard_normal(40) * np.exp(-np.arange(40) / 8.0)
    d = np.convolve(x, h)[: L * nb] + 0.01 * rng.standard_normal(L * nb)
    f = FastFreqLms(filter_len=L, mu=0.05)
    pa = np.ones(nb)
    e, w, W, P = run(f, x, d, pa, None)
    out = dict(a_x=x, a_d=d, a_p=pa, a_e=e, a_w=w, a_W=W, a_P=P, a_params=np.array([L, 1, 0.05, 0.9, 0, -1]))
    # (b) three channels, non-causal, per-bin p, fir_truncate=30 — the TDGSC canceller's configuration (TDGSC.py:37,105)
    L, nb, C = 256, 30, 3
    x = rng.standard_normal((L * nb, C)) * 0.2
    d = sum(np.convolve(x[:, c], rng.standard_normal(60) * np.exp(-np.arange(60) / 10.0))[: L * nb] for c in range(C))
    d = d + 0.01 * rng.standard_normal(L * nb)
    pb = rng.uniform(0, 1, (nb, L + 1))
    f = FastFreqLms(filter_len=L, n_channels=C, non_causal=True)
    e, w, W, P = run(f, x, d, pb, 30)
    out.update(b_x=x, b_d=d, b_p=pb, b_e=e, b_w=w, b_W=W, b_P=P, b_params=np.array([L, C, 0.01, 0.9, 1, 30]))
    # (c) coefficient-clamped blocking filter (FDGSC.py:71-81 configuration)
    L, nb = 256, 30
    x = rng.standard_normal(L * nb) * 0.2
    d = np.concatenate((np.zeros(L // 2), x))[: L * nb] * 0.8 + 0.02 * rng.standard_normal(L * nb)
    f = AdaptiveBlockingMatrixFilter(filter_len=L, mu=0.1, alpha=0.9, non_causal=False, constrain=True)
    pc = np.ones(nb)
    e, w, W, P = run(f, x, d, pc, None)
    out.update(c_x=x, c_d=d, c_p=pc, c_e=e, c_w=w, c_W=W, c_P=P, c_params=np.array([L, 1, 0.1, 0.9, 0, -1]))
    # (d) norm-limited canceller (FDGSC.py:83-91 configuration), scalar p per block; strong coupling so the limiter engages
    L, nb, C = 128, 40, 4
    x = rng.standard_normal((L * nb, C)) * 0.2
    d = sum(np.convolve(x[:, c], rng.standard_normal(30) * 0.6)[: L * nb] for c in range(C))
    f = AdaptiveInterferenceCancellation(filter_len=L, n_channels=C, mu=0.1, alpha=0.9, non_causal=False, constrain=True,
                                         weight_norm=True)
    pd_ = rng.uniform(0.2, 1.0, nb)
    e, w, W, P = run(f, x, d, pd_, None)
    nrm = np.sum(np.abs(W) ** 2) / f.n_fft / f.n_fft
    out.update(d_x=x, d_d=d, d_p=pd_, d_e=e, d_w=w, d_W=W, d_P=P, d_params=np.array([L, C, 0.1, 0.9, 0, -1]), d_final_norm=np.array(nrm))
    save("g14_fdaf", "FastFreqLms.update (FastFreqLms.py:204-245), AdaptiveBlockingMatrixFilter.update (gsc_bm.py:61-122), "
         "AdaptiveInterferenceCancellation.update (gsc_aic.py:53-108) block by block; params = [filter_len, n_channels, mu, alpha, "
         "non_causal, fir_truncate(-1 = None)]", **out)


def g14b_fdaf_two_path():
    """FastFreqLms(two_path=True) (FastFreqLms.py:94-104,162-176): foreground / background filters with the 3 dB transfer rule; the
    system changes half way so that transfers happen at the start and after the change."""
    from DistantSpeech.adaptivefilter.FastFreqLms import FastFreqLms
    rng = np.random.default_rng(142)
    out = {}
    for tag, L, nb, C in (("e", 128, 60, 1), ("f", 64, 50, 2)):
        x = rng.standard_normal((L * nb, C)) * 0.3
        d = np.zeros(L * nb)
        half = L * nb // 2
        for c in range(C):
            h1 = rng.standard_normal(40) * np.exp(-np.arange(40) / 8.0)
            h2 = rng.standard_normal(40) * np.exp(-np.arange(40) / 6.0)
            d[:half] += np.convolve(x[:, c], h1)[:half]
            d[half:] += np.convolve(x[:, c], h2)[half: L * nb]
        d += 0.01 * rng.standard_normal(L * nb)
        f = FastFreqLms(filter_len=L, mu=0.05, n_channels=C, two_path=True)
        e = np.zeros(L * nb)
        transfers = []
        for n in range(nb):
            fg0 = f.foreground.copy()
            en, w = f.update(x[n * L:(n + 1) * L] if C > 1 else x[n * L:(n + 1) * L, 0], d[n * L:(n + 1) * L])
            e[n * L:(n + 1) * L] = en[:, 0]
            transfers.append(int(not np.array_equal(fg0, f.foreground)))
        out.update({tag + "_x": x, tag + "_d": d, tag + "_e": e, tag + "_w": np.array(w), tag + "_W": np.array(f.W),
                    tag + "_F": np.array(f.foreground), tag + "_transfers": np.array(transfers), tag + "_params": np.array([L, C, 0.05, 0.9])})
    save("g14b_fdaf_two_path", "FastFreqLms(two_path=True).update FastFreqLms.py:94-104,162-176,204-245", **out)


def g15_tdgsc(x16):
    from DistantSpeech.beamformer.TDGSC import TDGSC
    x = x16.astype(np.float32) / 32768.0
    for name, xx, M, pf in (("rec1", x[:, : 256 * 150], 4, False), ("rec1_pf", x[:, : 256 * 150], 4, True),
                            ("synth_m6_pf", synth(151, 6, 256 * 80), 6, True)):
        mic = MicArray(arrayType="circular", r=0.032 if M == 4 else 0.05, M=M, n_fft=512)
        with contextlib.redirect_stdout(io.StringIO()):
            g = TDGSC(mic, frameLen=256, angle=[197, 0])
            out, p, obm = g.process(xx.T.astype(np.float64).copy(), postfilter=pf)
        save("g15_tdgsc_%s" % name, "TDGSC.process(postfilter=%s) TDGSC.py:110-175" % pf,
             x=(x16[:, : 256 * 150] if name.startswith("rec1") else xx), output=out, p=p, output_bm=obm.astype(np.float32),
             w=np.array(g.aic_filter.w), params=np.array([M, 256, int(pf)]), r=np.array(mic.r))


def g16_fdgsc(x16):
    from DistantSpeech.beamformer.FDGSC import FDGSC
    x = x16.astype(np.float32) / 32768.0
    # broadband burst after the SPP's 2L = 120 start-up blocks: drives mean(p[32:128]) over 0.8 (FDGSC.py:251-253)
    rng = np.random.default_rng(162)
    burst = rng.standard_normal((4, 256 * 180)) * 0.01
    s = rng.standard_normal(256 * 180) * 0.3
    s[: 256 * 135] = 0
    s[256 * 160:] = 0
    burst = (burst + s[None, :]).astype(np.float32)
    # the same with frameLen 64 (K = 65): np.mean(p_bm[32:128]) then averages the 33 bins that exist (FDGSC.py:248)
    rng = np.random.default_rng(163)
    burst64 = rng.standard_normal((4, 64 * 400)) * 0.01
    s = rng.standard_normal(64 * 400) * 0.3
    s[: 64 * 300] = 0
    s[64 * 360:] = 0
    burst64 = (burst64 + s[None, :]).astype(np.float32)
    for name, xx, M, pf, FL in (("rec1", x[:, : 256 * 150], 4, False, 256), ("rec1_pf", x[:, : 256 * 150], 4, True, 256),
                                ("synth_m6_pf", synth(161, 6, 256 * 80), 6, True, 256), ("burst", burst, 4, False, 256),
                                ("burst64", burst64, 4, False, 64)):
        mic = MicArray(arrayType="circular", r=0.032 if M == 4 else 0.05, M=M, n_fft=2 * FL)
        with contextlib.redirect_stdout(io.StringIO()):
            g = FDGSC(mic, frameLen=FL, angle=[197, 0])
            r = g.process(xx.T.astype(np.float64).copy(), postfilter=pf)
        save("g16_fdgsc_%s" % name, "FDGSC.process(postfilter=%s, dc_notch=True) FDGSC.py:201-317 (blocking-matrix mode 3), frameLen %d" % (pf, FL),
             x=(x16[:, : 256 * 150] if name.startswith("rec1") else xx), output=r[0], p=r[1], fix_output=r[2],
             fix_output_delayed=r[3], bm_output=r[4].astype(np.float32), aligned_output_delayed=r[6].astype(np.float32),
             w_aic=np.array(g.aic_filter.w).astype(np.float32), w_bm0=np.array(g.bm[0].w), params=np.array([M, FL, int(pf)]),
             r=np.array(mic.r))


def an101_int16():
    """[8, L] int16: the reference's 8-channel recording (example/test_audio/an101-mtms-arrA, 2.85 s), whole hops of 512."""
    files = sorted(glob.glob(os.path.join(_ref_shim.REFERENCE_ROOT, "example/test_audio/an101-mtms-arrA/*.wav")))
    assert len(files) == 8
    chans = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for f in files:
            sr, d = wavfile.read(f)
            assert sr == 16000 and d.dtype == np.int16
            chans.append(d)
    x = np.stack(chans)
    return x[:, : (x.shape[1] // 512) * 512]


SNAP_FRAMES = (1, 500, 1000)


def g17_long():
    """The reference's WHOLE test recording (rec1: 26.7 s = 1 670 hops of 256) through the three frame loops whose gate / VAD decisions
    accumulate over a long run: adaptivebeamfomer.process, GSC.process, SubbandGSC.process.  The input is stored once
    (int16, g17_rec1_full); each output fixture holds float32 samples (the ISTFT's own storage type, transform.py:243,359) and state
    snapshots at frames {1, 500, 1000, last}."""
    import DistantSpeech.beamformer.FDGSC as FD
    FD.DelayObj = object                                                                   # R9
    from DistantSpeech.beamformer.SubbandGSC import SubbandGSC
    x16 = rec1_int16(0.0, 1e9)
    x = x16.astype(np.float32) / 32768.0
    T = x.shape[1] // 256
    save("g17_rec1_full", "example/test_audio/rec1, all %d hops of 256 samples, 4 channels, int16" % T, x=x16)
    mic = MicArray(arrayType="circular", r=0.032, M=4, n_fft=512)                          # R2
    # -- adaptive MVDR
    ab = make_adaptive(mic, 512, 256)                                                      # R1
    ys, snap = [], {}
    with contextlib.redirect_stdout(io.StringIO()):
        for t in range(T):                                                                 # R3
            ys.append(np.atleast_1d(ab.process(x[:, t * 256:(t + 1) * 256].astype(np.float64), ANGLE, method=2)["data"]))
            if t in SNAP_FRAMES:
                snap["Rvv_t%d" % t] = ab.Rvv.astype(np.complex64)
                snap["p_t%d" % t] = ab.mcra.p.astype(np.float32)
    save("g17_adaptive_rec1_full", "adaptivebeamfomer.process(method=2) adaptivebeamformer.py:44-128 hop-by-hop over the whole recording; "
         "R1 R2 R3; angle=197deg; snapshots after frames %s and the last" % (SNAP_FRAMES,),
         y=np.concatenate(ys).astype(np.float32), Rvv=ab.Rvv, H=ab.H, mcra_p=ab.mcra.p, mcra_lambda_d=ab.mcra.lambda_d,
         params=np.array([4, 512, 256, 2]), r=np.array(mic.r), **snap)
    # -- GSC
    with contextlib.redirect_stdout(io.StringIO()):                                        # R5
        g = GSC(mic, frameLen=512, angle=[197, 0])
    ys, snap = [], {}
    with contextlib.redirect_stdout(io.StringIO()):
        for t in range(T):                                                                 # R3
            ys.append(np.atleast_1d(g.process(x[:, t * 256:(t + 1) * 256].astype(np.float64), ANGLE, method=2)["data"]))
            if t in SNAP_FRAMES:
                snap["G_t%d" % t] = g.G.astype(np.complex64)
                snap["spp_p_t%d" % t] = g.spp.p.astype(np.float32)
    save("g17_gsc_rec1_full", "GSC.process(method=2) GSC.py:174-294 hop-by-hop over the whole recording; R2 R3 R5; angle=197deg",
         y=np.concatenate(ys).astype(np.float32), G=g.G, spp_G=g.spp.G, spp_p=g.spp.p, params=np.array([4, 512, 256, 2]), r=np.array(mic.r), **snap)
    # -- SubbandGSC (one call over the whole recording, as example/run_GSC.py feeds it)
    with contextlib.redirect_stdout(io.StringIO()):
        sg = SubbandGSC(mic, frameLen=256, angle=[197, 0])
        with np.errstate(all="ignore"):
            out, fix, bm, p, al = sg.process(x.astype(np.float64).copy())
    save("g17_subbandgsc_rec1_full", "SubbandGSC.process(postfilter=False) SubbandGSC.py:170-262 over the whole recording; R9",
         output=out.astype(np.float32), fix_output=fix.astype(np.float32), bm_output=bm[:, ::4].astype(np.float32), p=p.astype(np.float32),
         params=np.array([4, 256, 0]), r=np.array(mic.r))


def g18_an101():
    """The reference's real 8-channel recording (an101-mtms-arrA, 8 x 2.85 s; example/run_postfilter.py builds its array as
    MicArray('linear', r=0.032, M=8)) through adaptivebeamfomer at 1024 / 512 and through the patched Wpe (R6, R7) on the same grid."""
    from DistantSpeech.dereverberation import awpe
    x16 = an101_int16()
    x = x16.astype(np.float32) / 32768.0
    M, nfft, hop = 8, 1024, 512
    T = x.shape[1] // hop
    mic = MicArray(arrayType="linear", r=0.032, M=M, n_fft=nfft)                           # R2
    ab = make_adaptive(mic, nfft, hop)                                                     # R1
    ys = []
    with contextlib.redirect_stdout(io.StringIO()):
        for t in range(T):                                                                 # R3
            ys.append(np.atleast_1d(ab.process(x[:, t * hop:(t + 1) * hop].astype(np.float64), ANGLE, method=2)["data"]))
    save("g18_adaptive_an101", "adaptivebeamfomer.process(method=2) adaptivebeamformer.py:44-128 hop-by-hop on an101-mtms-arrA "
         "(8 channels, 1024/512, MicArray linear r=0.032 as example/run_postfilter.py); R1 R2 R3; angle=197deg",
         x=x16, y=np.concatenate(ys), Rvv=ab.Rvv, H=ab.H, mcra_p=ab.mcra.p, params=np.array([M, nfft, hop, 2]), r=np.array(mic.r))
    awpe.Subband = Transform                                                               # R6

    def check_input_data(self, xd, xx):                                                    # R7
        self.return_td = True
        return np.squeeze(self.transform_x.analysis(xd)), np.squeeze(self.transform_d.analysis(xx))

    awpe.Wpe.check_input_data = check_input_data
    C, N, D = 8, 2, 4
    with contextlib.redirect_stdout(io.StringIO()):
        wpe = awpe.Wpe(filter_len=N, delay=D, channels=C, num_bands=nfft, hop_length=hop)
    outs = []
    xt = x.T.astype(np.float64)
    for n in range(T):
        out, _ = wpe.update(xt[n * hop:(n + 1) * hop])
        outs.append(np.atleast_1d(out))
    save("g18_wpe_an101", "Wpe.update awpe.py:129-192 hop-by-hop on an101-mtms-arrA, 8 channels x 2 taps, delay 4, STFT grid 1024/512 "
         "(BASELINE config 4's shape); R6 R7 (PARITY UNPINNED by the reference as shipped)",
         y=np.concatenate(outs), W=wpe.W.astype(np.complex64), var=wpe.var, params=np.array([C, N, D, nfft, hop]))


def g19_gev(x16):
    """The notebook's GEV flow (example/mvdr.ipynb: get_gev_vector -> phase_correction -> blind_analytic_normalization -> output) and the
    free compute_pmwf_weight, on PSD matrices of the reference's recording: target = average of all frames, noise = average of the
    frames whose power is below the median (plus 1e-6 of its trace on the diagonal)."""
    from DistantSpeech.beamformer.beamformer import get_gev_vector, phase_correction, blind_analytic_normalization
    x = x16.astype(np.float32) / 32768.0
    M = 4
    tr = Transform(n_fft=512, hop_length=256, channel=M)
    D = tr.stft(x.T.astype(np.float64))                                                    # [K, T, M]
    pw = np.mean(np.abs(D) ** 2, axis=(0, 2))
    quiet = pw < np.median(pw)
    Phi_yy = np.einsum("ktm,ktn->kmn", D, D.conj()) / D.shape[1]
    Phi_vv = np.einsum("ktm,ktn->kmn", D[:, quiet], D[:, quiet].conj()) / quiet.sum()
    Phi_vv = Phi_vv + 1e-6 * np.real(np.trace(Phi_vv, axis1=1, axis2=2))[:, None, None] * np.eye(M)[None]
    Phi_xx = Phi_yy - Phi_vv
    W = get_gev_vector(Phi_xx, Phi_vv)
    Wp = phase_correction(W)
    Wb = blind_analytic_normalization(Wp, Phi_vv)
    Yout = np.einsum("inj,ij->in", D, Wb.conj())
    y = tr.istft(Yout[:, :, None])
    xi = np.real(np.trace(np.linalg.inv(Phi_vv) @ Phi_xx, axis1=1, axis2=2))
    Rvv_inv = np.linalg.inv(Phi_vv)
    u = np.zeros((257, M, 1)); u[:, 0, 0] = 1                                              # R10: compute_pmwf_weight's body with channels = M
    w_pmwf = {b: (Rvv_inv @ Phi_xx @ u).squeeze() / (b + xi[:, None]) for b in (1, 10)}
    save("g19_gev", "get_gev_vector / phase_correction / blind_analytic_normalization beamformer/beamformer.py:34-97 as example/mvdr.ipynb "
         "chains them, and compute_pmwf_weight :100-130 (R10) on PSD matrices of rec1", x=x16, Phi_xx=Phi_xx, Phi_vv=Phi_vv, W_gev=W, W_pc=Wp,
         W_ban=Wb, Yout=Yout.astype(np.complex64), y=np.asarray(y), xi=xi, w_pmwf_b1=w_pmwf[1], w_pmwf_b10=w_pmwf[10])


def g20_odd_m():
    """Odd channel counts through the beamformer objects: the adaptive MVDR (method 2) and the GSC hop by hop with 3 and 5 microphones.
    (7 cannot be built: MicArray.gamma = arange(0, 360, int(360 / M)) has 8 entries for M = 7, MicArray.py:33, and the steering vector
    beamformer.py:267-289 takes one entry per element of gamma for linear arrays too — getweights fails with a shape mismatch.)"""
    for name, M, atype, seed in (("m3", 3, "circular", 71), ("m5", 5, "circular", 72)):
        x = synth(seed, M, 256 * 50)
        mic = MicArray(arrayType=atype, r=0.05, M=M, n_fft=512)                            # R2
        ab = make_adaptive(mic, 512, 256)                                                  # R1
        ys = []
        with contextlib.redirect_stdout(io.StringIO()):
            for t in range(x.shape[1] // 256):                                             # R3
                ys.append(np.atleast_1d(ab.process(x[:, t * 256:(t + 1) * 256].astype(np.float64), ANGLE, method=2)["data"]))
        save("g4_adaptive_synth_%s" % name, "adaptivebeamfomer.process(method=2) adaptivebeamformer.py:44-128 hop-by-hop, %d microphones (%s); R1 R2 R3; "
             "angle=197deg" % (M, atype), x=x, y=np.concatenate(ys), Rvv=ab.Rvv, Rvv_inv=ab.Rvv_inv, Ryy=ab.Ryy, H=ab.H, mcra_p=ab.mcra.p,
             mcra_lambda_d=ab.mcra.lambda_d, params=np.array([M, 512, 256, 2]), r=np.array(mic.r))
    for name, M, seed in (("m3", 3, 74), ("m5", 5, 75)):
        xx = synth(seed, M, 256 * 50) * np.float32(0.1)
        mic = MicArray(arrayType="circular", r=0.05, M=M, n_fft=512)                       # R2
        with contextlib.redirect_stdout(io.StringIO()):                                    # R5
            g = GSC(mic, frameLen=512, angle=[197, 0])
        ys = []
        with contextlib.redirect_stdout(io.StringIO()):
            for t in range(xx.shape[1] // 256):                                            # R3
                ys.append(np.atleast_1d(g.process(xx[:, t * 256:(t + 1) * 256].astype(np.float64), ANGLE, method=2)["data"]))
        save("g6_gsc_synth_%s" % name, "GSC.process(method=2) GSC.py:174-294 hop-by-hop, %d microphones; R2 R3 R5; angle=197deg" % M,
             x=xx, y=np.concatenate(ys), G=g.G, spp_G=g.spp.G, spp_p=g.spp.p, omlsa_G=g.omlsa_multi.G, omlsa_p=g.omlsa_multi.p,
             omlsa_lambda_d=np.asarray(g.omlsa_multi.lambda_d), mcra_p=g.mcra.p, params=np.array([M, 512, 256, 2]), r=np.array(mic.r))


def g21_wpe_wide():
    """The reference's maintained use of Wpe: Wpe(channels=4, mu=1e-4, forgetting_factor=0.998, filter_len=20, delay=4, num_bands=256,
    hop_length=64) driven one hop (64 samples) per update() call (example/wpe.ipynb cell 2) — on 4 s of the reference's own 4-channel
    recording rec1 (the notebook's input files are not in the repository); and 8 channels x 10 taps on the 1024 / 512 grid (SURVEY 8d's
    sizing of BASELINE config 4) on the reference's 8-channel recording an101.  Patched reference: R6, R7."""
    from DistantSpeech.dereverberation import awpe
    awpe.Subband = Transform                                                               # R6

    def check_input_data(self, xd, xx):                                                    # R7
        self.return_td = True
        return np.squeeze(self.transform_x.analysis(xd)), np.squeeze(self.transform_d.analysis(xx))

    awpe.Wpe.check_input_data = check_input_data
    for name, x16, C, N, D, nb, hop, secs in (("nb_c4n20", rec1_int16(3.0, 4.0), 4, 20, 4, 256, 64, 4.0),
                                              ("c8n10", an101_int16(), 8, 10, 4, 1024, 512, 2.8)):
        T = min(int(secs * 16000) // hop, x16.shape[1] // hop)
        x16 = np.ascontiguousarray(x16[:, : T * hop])
        xt = (x16.astype(np.float32) / 32768.0).T.astype(np.float64)
        with contextlib.redirect_stdout(io.StringIO()):
            wpe = awpe.Wpe(channels=C, mu=1e-4, forgetting_factor=0.998, filter_len=N, delay=D, num_bands=nb, hop_length=hop)
        outs, W_mid = [], None
        for n in range(T):
            out, _ = wpe.update(xt[n * hop:(n + 1) * hop])
            outs.append(np.atleast_1d(out))
            if n == T // 2 - 1:
                W_mid = wpe.W[::8].astype(np.complex64)
        kk = np.arange(0, nb // 2 + 1, 8)
        kp = kk[:: max(1, len(kk) // 8)]                                                   # P (80 x 80 per bin) at a handful of bins
        save("g21_wpe_%s" % name, "Wpe.update awpe.py:129-192 one hop per call, Wpe(channels=%d, filter_len=%d, delay=%d, num_bands=%d, hop_length=%d, "
             "forgetting_factor=0.998) on the STFT grid (example/wpe.ipynb cell 2's operating point for nb_c4n20); W at every 8th bin, P at bins_P; "
             "R6 R7 (PARITY UNPINNED by the reference as shipped)" % (C, N, D, nb, hop),
             x=x16, y=np.concatenate(outs).astype(np.float32), W=wpe.W[kk].astype(np.complex64), W_mid=W_mid, P=wpe.P[kp].astype(np.complex64),
             var=wpe.var, bins=kk, bins_P=kp, params=np.array([C, N, D, nb, hop]))


def g23_mvdr_postfilter(x16):
    """MVDR + McMcra post-filter in one pass (BASELINE.json north_star's target workload).  No reference class composes the two;
    this drives the REFERENCE's own objects frame by frame and composes them the way GSC.process composes its beamformer with `spp`
    (GSC.py:225 `self.spp.estimation(Z)`, GSC.py:286 `Y * self.spp.G`): the weights are adaptivebeamfomer's own `H` after its
    process() call of the hop, the gain is McMcra's own `G` for the same input frame, analysis / synthesis are Transform's."""
    cases = [("rec1", x16.astype(np.float32) / 32768.0, 4, 512, 256),
             ("synth", synth(5, 4, 256 * 90), 4, 512, 256),
             ("synth_m6", synth(7, 6, 256 * 60), 6, 512, 256),
             ("synth_m2_256", synth(11, 2, 128 * 80), 2, 256, 128),
             ("synth_m8_1024", synth(9, 8, 512 * 30), 8, 1024, 512),          # round 6: the shapes of BASELINE config 4 (8 microphones, 1024 points)
             ("synth_m6_1024", synth(12, 6, 512 * 30), 6, 1024, 512)]
    only_new = os.environ.get("G23_ONLY_NEW") == "1"                           # (regenerating the round-6 cases alone)
    for name, x, M, nfft, hop in cases:
        if only_new and name not in ("synth_m8_1024", "synth_m6_1024"):
            continue
        mic = MicArray(arrayType="circular", r=0.032 if M == 4 else 0.05, M=M, n_fft=nfft)   # R2
        ab = make_adaptive(mic, nfft, hop)                                                 # R1
        tr_in = Transform(n_fft=nfft, hop_length=hop, channel=M)
        tr_out = Transform(n_fft=nfft, hop_length=hop, channel=1)
        spp = McMcra(nfft=nfft, channels=M)
        T = x.shape[1] // hop
        K = nfft // 2 + 1
        ys, y_mvdr = [], []
        G = np.zeros((T, K)); P = np.zeros((T, K))
        with contextlib.redirect_stdout(io.StringIO()):
            for t in range(T):                                                             # R3
                xh = x[:, t * hop:(t + 1) * hop].astype(np.float64)
                out = ab.process(xh, ANGLE, method=2)                                      # adaptivebeamformer.py:44-128: updates ab.H
                y_mvdr.append(np.atleast_1d(out["data"]))
                Z = tr_in.stft(xh.T)[:, 0, :]                                              # [K, M], the frame ab.process analysed
                spp.estimation(Z)                                                          # GSC.py:225
                Y = np.sum(np.conj(np.asarray(ab.H)).T * Z, axis=1) * spp.G                # adaptivebeamformer.py:119-120, GSC.py:286
                ys.append(np.atleast_1d(tr_out.istft(Y[:, None, None])))
                G[t], P[t] = spp.G, spp.p
        save("g23_mvdr_pf_%s" % name,
             "composition of reference objects, per hop: adaptivebeamfomer.process(method=2) -> H; Transform.stft -> Z; McMcra.estimation(Z) -> G; "
             "Y = (H^H Z) G (adaptivebeamformer.py:119-120, GSC.py:225,286); Transform.istft.  R1 R2 R3; angle=197deg.  PARITY of the COMPOSITION "
             "is defined by this script (no reference class composes them); every object in it is the reference's",
             x=(x16 if name == "rec1" else x), y=np.concatenate(ys), y_mvdr=np.concatenate(y_mvdr), G=G[::4], p=P[::4], G_last=G[-1], p_last=P[-1],
             Rvv=ab.Rvv, Phi_vv=np.moveaxis(spp.Phi_vv, 2, 0), Phi_yy=np.moveaxis(spp.Phi_yy, 2, 0), mcra_p=ab.mcra.p,
             params=np.array([M, nfft, hop, 2]), r=np.array(mic.r))


def g24_estpos_reth(x16):
    """adaptivebeamfomer with `estPos` set (adaptivebeamformer.py:30,90-93: Rvv from the first estPos (frame, bin) slots after a restart; the look
    direction changes at hop 25, which restarts the count, :70-79) and process(retH=True) (:124-126, beamformer.py:536-553) on the last hop;
    GSC.process(retH=True) (GSC.py:290-292: its H is the constructor's ones / M)."""
    M, nfft, hop, K = 4, 512, 256, 257
    x = synth(11, M, hop * 40)
    angle2 = np.array([90, 0]) / 180 * np.pi
    for name, est, method in (("estpos", 12 * K + 100, 2), ("estpos_whole_frames", 8 * K, 2), ("vad_tfgsc", None, 3), ("vad_ds", None, 1)):
        mic = MicArray(arrayType="circular", r=0.032, M=M, n_fft=nfft)                       # R2
        ab = make_adaptive(mic, nfft, hop)                                                 # R1
        ab.estPos = est
        ys, bp = [], None
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(divide="ignore"):
            for t in range(40):                                                            # R3
                ang = ANGLE if t < 25 else angle2
                out = ab.process(x[:, t * hop:(t + 1) * hop].astype(np.float64), ang, method=method, retH=(t == 39))
                ys.append(np.atleast_1d(out["data"]))
                bp = out["beampattern"]
        save("g24_adaptive_%s" % name, "adaptivebeamfomer.process(method=%d, retH on the last hop), estPos=%r, look direction 197 -> 90 deg at hop 25; "
             "adaptivebeamformer.py:44-128 hop-by-hop; R1 R2 R3" % (method, est),
             x=x, y=np.concatenate(ys), Rvv=ab.Rvv, H=ab.H, beampattern=bp[::6].astype(np.float32), bp_az=np.arange(0, 360, 6), est_pos=np.array(-1 if est is None else est),
             frame_count=np.array(ab.frameCount), params=np.array([M, nfft, hop, method]), r=np.array(mic.r))
    mic = MicArray(arrayType="circular", r=0.032, M=M, n_fft=nfft)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(divide="ignore"):
        g = GSC(mic, frameLen=nfft, angle=[197, 0])
        ys = []
        for t in range(12):
            out = g.process(x[:, t * hop:(t + 1) * hop].astype(np.float64), ANGLE, method=2, retH=(t == 11))
            ys.append(np.atleast_1d(out["data"]))
    save("g24_gsc_reth", "GSC.process(method=2, retH on the last hop) GSC.py:174-294 hop-by-hop; R2 R3",
         x=x[:, :12 * hop], y=np.concatenate(ys), beampattern=out["beampattern"][::6].astype(np.float32), bp_az=np.arange(0, 360, 6),
         params=np.array([M, nfft, hop, 2]), r=np.array(mic.r))


def main():
    only = set(sys.argv[1:])     # e.g. `make_golden.py g6` regenerates one family

    def want(tag):
        return not only or tag in only

    x16 = rec1_int16(3.0, 3.0)
    if want("g1"): g1_transform()
    if want("g1b"): g1b_transform_window()
    if want("g1c"): g1c_transform_quarter_hop()
    if want("g2"): g2_weights()
    if want("g2b"): g2b_fixed(x16)
    if want("g3"): g3_mcra(x16)
    if want("g4"): g4_adaptive(x16)
    if want("g5"): g5_mcmcra(x16)
    if want("g6"): g6_gsc(x16)
    if want("g7"): g7_omlsa()
    if want("g8"): g8_subband()
    if want("g9"): g9_mcsppbase(x16)
    if want("g10"): g10_wpe()
    if want("g11"): g11_mcspp(x16)
    if want("g11b"): g11b_mcspp_an101()
    if want("g12"): g12_subbandgsc(x16)
    if want("g13"): g13_tdfilters()
    if want("g14"): g14_fdaf()
    if want("g14b"): g14b_fdaf_two_path()
    if want("g15"): g15_tdgsc(x16)
    if want("g16"): g16_fdgsc(x16)
    if want("g17"): g17_long()
    if want("g18"): g18_an101()
    if want("g19"): g19_gev(x16)
    if want("g20"): g20_odd_m()
    if want("g21"): g21_wpe_wide()
    if want("g22"): g22_subbandgsc_postfilter(x16)
    if want("g23"): g23_mvdr_postfilter(x16)
    if want("g24"): g24_estpos_reth(x16)


if __name__ == "__main__":
    main()
