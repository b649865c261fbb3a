"""Pin the CPU oracle (oracle/ds_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest

from oracle import ds_oracle as O
from conftest import rms

ANGLE = np.array([197, 0]) / 180 * np.pi


def _mic(M, nfft, r=None, atype="circular"):
    return O.OracleMicArray(arrayType=atype, r=(0.032 if M == 4 else 0.05) if r is None else r, M=M, n_fft=nfft)


@pytest.mark.parametrize("name", ["g1_transform_512_256_4", "g1_transform_1024_512_2", "g1_transform_256_128_1",
                                  "g1c_transform_512_128_2", "g1c_transform_256_64_4", "g1c_transform_1024_256_1", "g1c_transform_512_128_5"])
def test_transform(golden, name):
    g = golden(name)
    nfft, hop, M = [int(v) for v in g["params"]]
    x = g["x"]
    t = O.OracleTransform(channel=M, n_fft=nfft, hop_length=hop)
    Y = t.stft(x)
    assert np.array_equal(Y.astype(np.complex64), g["Y"])          # bit-exact complex64 STFT
    y = np.asarray(t.istft(Y)).reshape(x.shape[0], -1)
    assert np.max(np.abs(y - g["y"])) < 1e-7
    # chunked == one-shot
    t2 = O.OracleTransform(channel=M, n_fft=nfft, hop_length=hop)
    cuts = [0, hop, 3 * hop, 10 * hop, x.shape[0]] if name.startswith("g1c") else [0, hop, 4 * hop, x.shape[0]]   # as make_golden.py cut them
    ys = [np.asarray(t2.istft(t2.stft(x[a:b]))).reshape(b - a, -1) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.max(np.abs(np.concatenate(ys) - g["y_chunk"])) < 1e-7


@pytest.mark.parametrize("name,atype", [("g2_weights_circular_M4_512", "circular"), ("g2_weights_linear_M6_512", "linear"),
                                        ("g2_weights_circular_M8_1024", "circular")])
def test_weights(golden, name, atype):
    g = golden(name)
    M, nfft, az, el = g["params"]
    M, nfft = int(M), int(nfft)
    mic = _mic(M, nfft, r=float(g["r"]), atype=atype)
    assert np.allclose(mic.mic_loc, g["mic_loc"], atol=1e-15)
    assert np.allclose(O.steering_from_doa(mic, nfft, (az, el)), g["a0"], atol=1e-12)
    assert np.allclose(O.gen_noise_msc(mic, nfft), g["Fvv"], atol=1e-12)
    assert np.allclose(O.fixed_weights(mic, nfft, (az, el), "DS"), g["Wds"], atol=1e-12)
    assert np.allclose(O.fixed_weights(mic, nfft, (az, el), "SD"), g["Wsd"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("wt", ["DS", "SD"])
def test_fixed_beamformer(golden, wt):
    g = golden("g2b_fixed_" + wt)
    x = g["x"].astype(np.float32) / 32768.0
    fb = O.OracleFixedBeamformer(_mic(4, 512), frameLen=512, angle=(197, 0), weightType=wt)
    assert np.allclose(fb.W, g["W"], rtol=1e-9, atol=1e-9)
    y = fb.process(x.T)
    assert rms(y - g["y"]) < 1e-7 * max(rms(g["y"]), 1e-3)


@pytest.mark.parametrize("L", [15, 10])
def test_mcra(golden, L):
    g = golden("g3_mcra_L%d" % L)
    est = O.OracleMCRA(nfft=512, L=L)
    P = g["P"]
    for n in range(P.shape[0]):
        est.estimation(P[n])
        assert np.allclose(est.lambda_d, g["lambda_d"][n], rtol=1e-12, atol=1e-18), n
        assert np.allclose(est.p, g["p"][n], rtol=1e-12, atol=1e-15), n
        if n % 8 == 0:
            assert np.allclose(est.S, g["S"][n // 8], rtol=1e-12, atol=1e-18)
            assert np.allclose(est.Smin, g["Smin"][n // 8], rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("name", ["rec1", "synth", "synth_ds", "synth_src", "synth_tfgsc", "synth_m6", "synth_m8_1024", "synth_m3", "synth_m5"])
def test_adaptive_mvdr(golden, name):
    g = golden("g4_adaptive_" + name)
    M, nfft, hop, method = [int(v) for v in g["params"]]
    x = g["x"]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    ab = O.OracleAdaptiveMVDR(_mic(M, nfft, r=float(g["r"])), frameLen=nfft, hop=hop, nfft=nfft)
    y = ab.process(x, ANGLE, method=method)
    ref = g["y"]
    assert y.shape == ref.shape
    assert rms(y - ref) < 1e-7 * max(rms(ref), 1e-3), rms(y - ref)
    assert np.allclose(ab.Rvv, g["Rvv"], rtol=1e-9, atol=1e-14)
    assert np.allclose(ab.Ryy, g["Ryy"], rtol=1e-9, atol=1e-14)
    assert np.allclose(ab.mcra.p, g["mcra_p"], rtol=1e-10, atol=1e-14)
    if method == 2:
        assert np.allclose(ab.H, g["H"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", ["rec1", "synth", "synth_m6", "synth_m2_256", "synth_m8_1024", "synth_m6_1024"])
def test_mvdr_postfilter_one_pass(golden, name):
    """G23: MVDR + McMcra gain in one pass, composed from the reference's own objects (make_golden.py g23)."""
    g = golden("g23_mvdr_pf_" + name)
    M, nfft, hop, method = [int(v) for v in g["params"]]
    x = g["x"]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    o = O.OracleMvdrPostfilter(_mic(M, nfft, r=float(g["r"])), nfft=nfft, hop=hop)
    y = o.process(x, ANGLE, method=method)
    ref = g["y"]
    assert y.shape == ref.shape
    assert rms(y - ref) < 1e-7 * max(rms(ref), 1e-3), rms(y - ref)
    assert np.allclose(o.mvdr.Rvv, g["Rvv"], rtol=1e-9, atol=1e-14)
    assert np.allclose(o.spp.G, g["G_last"], rtol=1e-6, atol=1e-9)
    assert np.allclose(o.spp.p, g["p_last"], rtol=1e-6, atol=1e-9)
    assert np.allclose(o.spp.Phi_vv, g["Phi_vv"], rtol=1e-9, atol=1e-14)
    # the un-gained beamformer output of the same run is the plain adaptive MVDR's
    y0 = O.OracleAdaptiveMVDR(_mic(M, nfft, r=float(g["r"])), frameLen=nfft, hop=hop, nfft=nfft).process(x, ANGLE, method=method)
    assert rms(y0 - g["y_mvdr"]) < 1e-7 * max(rms(g["y_mvdr"]), 1e-3)


@pytest.mark.parametrize("name", ["estpos", "estpos_whole_frames", "vad_tfgsc", "vad_ds"])
def test_adaptive_estpos_and_beampattern(golden, name):
    """G24: `estPos` (adaptivebeamformer.py:30,90-93: Rvv from the first estPos (frame, bin) slots after a restart; the look direction changes at
    hop 25, which restarts the count, :70-79) and process(retH=True)'s beampattern (:124-126, beamformer.py:536-553)."""
    g = golden("g24_adaptive_" + name)
    M, nfft, hop, method = [int(v) for v in g["params"]]
    x = g["x"]
    ab = O.OracleAdaptiveMVDR(_mic(M, nfft, r=float(g["r"])), frameLen=nfft, hop=hop, nfft=nfft)
    ab.estPos = None if int(g["est_pos"]) < 0 else int(g["est_pos"])
    angle2 = np.array([90, 0]) / 180 * np.pi
    y = np.concatenate([ab.process(x[:, t * hop:(t + 1) * hop], ANGLE if t < 25 else angle2, method=method) for t in range(40)])
    assert rms(y - g["y"]) < 1e-7 * max(rms(g["y"]), 1e-3)
    assert np.allclose(ab.Rvv, g["Rvv"], rtol=1e-9, atol=1e-14)
    assert ab.frameCount == int(g["frame_count"])
    bp = ab.beampattern(ab.omega, ab.H)[g["bp_az"]]
    ok = np.isfinite(g["beampattern"])
    assert np.allclose(bp[ok], g["beampattern"][ok], rtol=0, atol=2e-4)              # dB; the fixture holds float32


def test_adaptive_mvdr_chunk_invariance(golden):
    g = golden("g4_adaptive_synth")
    x = g["x"][:, : 256 * 40]
    mic = _mic(4, 512)
    a = O.OracleAdaptiveMVDR(mic, 512).process(x, ANGLE)
    b_ = O.OracleAdaptiveMVDR(mic, 512)
    b = np.concatenate([b_.process(x[:, : 256 * 7], ANGLE), b_.process(x[:, 256 * 7:], ANGLE)])
    assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["rec1", "synth_m6"])
def test_mcmcra(golden, name):
    g = golden("g5_mcmcra_" + name)
    M, nfft, hop = [int(v) for v in g["params"]]
    x = g["x"]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    D = O.OracleTransform(channel=M, n_fft=nfft, hop_length=hop).stft(x.T)
    est = O.OracleMcMcra(nfft=nfft, channels=M)
    for n in range(D.shape[1]):
        est.estimation(D[:, n, :])
        assert np.allclose(est.p, g["p"][n], rtol=1e-6, atol=1e-9), n
        assert np.allclose(est.G, g["G"][n], rtol=1e-6, atol=1e-9), n
        if n % 8 == 0:
            assert np.allclose(est.xi, g["xi"][n // 8], rtol=1e-6, atol=1e-9)
            assert np.allclose(est.gamma, g["gamma"][n // 8], rtol=1e-6, atol=1e-9)
    assert np.allclose(est.Phi_vv, g["Phi_vv"], rtol=1e-9, atol=1e-14)
    assert np.allclose(est.Phi_yy, g["Phi_yy"], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("name", ["rec1", "synth_m6", "synth_m4", "synth_m0", "synth_m3", "synth_m5"])
def test_gsc(golden, name):
    g = golden("g6_gsc_" + name)
    M, nfft, hop, method = [int(v) for v in g["params"]]
    x = g["x"]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    gsc = O.OracleGSC(_mic(M, nfft, r=float(g["r"])), frameLen=nfft)
    y = gsc.process(x, ANGLE, method=method)
    ref = g["y"]
    assert rms(y - ref) < 1e-7 * max(rms(ref), 1e-3), rms(y - ref)
    assert np.allclose(gsc.spp.p, g["spp_p"], rtol=1e-6, atol=1e-9)
    assert np.allclose(gsc.spp.G, g["spp_G"], rtol=1e-6, atol=1e-9)
    if method != 0:
        assert np.allclose(gsc.G, g["G"], rtol=1e-7, atol=1e-10)
        assert np.allclose(gsc.omlsa_multi.G, g["omlsa_G"], rtol=1e-6, atol=1e-9)
        assert np.allclose(gsc.omlsa_multi.p, g["omlsa_p"], rtol=1e-6, atol=1e-9)
    assert np.allclose(gsc.mcra.p, g["mcra_p"], rtol=1e-10, atol=1e-14)


def test_omlsa(golden):
    g = golden("g7_omlsa")
    est = O.OracleOmlsaMulti(nfft=512, M=4, cal_weights=True)
    for n in range(g["y"].shape[0]):
        est.estimation(g["y"][n], g["u"][n])
        assert np.allclose(est.G, g["G"][n], rtol=1e-9, atol=1e-12), n
        assert np.allclose(est.p, g["p"][n], rtol=1e-9, atol=1e-12), n
        assert np.allclose(est.lambda_d, g["lambda_d"][n], rtol=1e-9, atol=1e-15), n
        if n % 8 == 0:
            assert np.allclose(est.xi_hat, g["xi_hat"][n // 8], rtol=1e-9, atol=1e-12)
            assert np.allclose(est.q_hat, g["q_hat"][n // 8], rtol=1e-9, atol=1e-12)


def test_subband(golden):
    g = golden("g8_subband")
    lms, rls = O.OracleSubbandLMS(2, 512, mu=0.1), O.OracleSubbandRLS(2, 512)
    mc = O.OracleSubbandLmsMc(2, 512, channel=g["xm"].shape[2], mu=0.1)
    for n in range(g["x"].shape[0]):
        e, _ = lms.update(g["x"][n], g["d"][n], p=g["p"][n])
        assert np.allclose(e, g["e_lms"][n], rtol=1e-10, atol=1e-12)
        e, _ = rls.update(g["x"][n], g["d"][n])
        assert np.allclose(e, g["e_rls"][n], rtol=1e-8, atol=1e-10)
        e, _ = mc.update(g["xm"][n], g["dm"][n], p=g["p"][n])
        assert np.allclose(e, g["e_mc"][n], rtol=1e-10, atol=1e-12)
    assert np.allclose(lms.W, g["W_lms"], rtol=1e-9, atol=1e-12)
    assert np.allclose(rls.W, g["W_rls"], rtol=1e-7, atol=1e-10)
    assert np.allclose(rls.P, g["P_rls"], rtol=1e-6, atol=1e-8)
    assert np.allclose(mc.W, g["W_mc"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["rec1", "synth_m6"])
def test_mcsppbase(golden, name):
    g = golden("g9_mcsppbase_" + name)
    M, nfft, hop = [int(v) for v in g["params"]]
    x = g["x"]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    D = O.OracleTransform(channel=M, n_fft=nfft, hop_length=hop).stft(x.T)
    est = O.OracleMcSppBase(nfft=nfft, channels=M)
    for n in range(D.shape[1]):
        est.estimation(D[:, n, :])
        assert np.allclose(est.p, g["p"][n], rtol=1e-6, atol=1e-9), n
        if n % 8 == 0:
            assert np.allclose(est.xi, g["xi"][n // 8], rtol=1e-6, atol=1e-9)
    assert np.allclose(est.w, g["w_last"], rtol=1e-5, atol=1e-8)
    assert np.allclose(est.Phi_vv, g["Phi_vv"], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("name", ["c4n2", "c2n3"])
def test_wpe(golden, name):
    """parity pinned only against the *patched* reference (make_golden.py R6, R7): Wpe does not run as shipped."""
    g = golden("g10_wpe_" + name)
    C, N, D, nb, hop = [int(v) for v in g["params"]]
    wpe = O.OracleWpe(channels=C, filter_len=N, num_bands=nb, delay=D, hop_length=hop)
    x = g["x"]
    y = np.concatenate([wpe.update(x[n:n + hop])[0] for n in range(0, x.shape[0], hop)])
    assert rms(y - g["y"]) < 1e-7 * max(rms(g["y"]), 1e-3)
    assert np.allclose(wpe.W, g["W"], rtol=1e-6, atol=1e-9)
    assert np.allclose(wpe.P, g["P"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("name", ["rec1", "synth_m6", "rec1_repeat"])
def test_mcspp_notebook_flow(golden, name):
    """McSpp + steering + compute_mvdr_weight, driven like example/mvdr.ipynb cell 4."""
    g = golden("g11_mcspp_" + name)
    M, nfft, hop = [int(v) for v in g["params"]]
    x = g["x"]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    D = O.OracleTransform(channel=M, n_fft=nfft, hop_length=hop).stft(x.T)
    est = O.OracleMcSpp(nfft, M)
    T = D.shape[1]
    Y = np.zeros((T, nfft // 2 + 1), dtype=complex)
    with np.errstate(all="ignore"):
        for n in range(T):
            p = est.estimation(D[:, n, :], repeat=name.endswith("_repeat"))
            assert np.allclose(p, g["p"][n], rtol=1e-7, atol=1e-10), n
            w = O.compute_mvdr_weight(O.steering(est.Phi_xx), est.Phi_vv_inv)
            Y[n] = np.einsum("ij,ij->i", w.conj(), D[:, n, :])
    assert rms(Y - g["Yout"]) < 1e-6 * rms(g["Yout"])
    assert np.allclose(est.Phi_vv_inv, g["Phi_vv_inv"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["rec1", "synth_m6", "synth_m6_rls"])
def test_subband_gsc(golden, name):
    g = golden("g12_subbandgsc_" + name)
    M, FL, rls = [int(v) for v in g["params"]]
    x = g["x"]
    if x.dtype == np.int16:
        x = x.astype(np.float32) / 32768.0
    sg = O.OracleSubbandGSC(_mic(M, 512, r=float(g["r"])), FL, (197, 0), rls_bm=bool(rls))
    assert np.allclose(sg.time_alignment.delay_filter, g["delay_filter"], atol=1e-14)
    out, fix, bm, p, al = sg.process(x)
    assert rms(out - g["output"]) < 1e-6 * rms(g["output"])
    assert rms(bm - g["bm_output"]) < 1e-6 * rms(g["bm_output"])
    assert np.max(np.abs(p - g["p"])) < 1e-6


@pytest.mark.parametrize("name", ["rec1_1", "rec1_5", "rec1_1_lvl1"])
def test_subband_gsc_postfilter_branch(golden, name):
    """SubbandGSC.process(postfilter=True) (SubbandGSC.py:236-249): same five results; the branch's trace is the object's omlsa_multi —
    driven one block per call and five (where every block re-analyses the whole bm_output array of the call and takes its frame 0)."""
    g = golden("g22_subbandgsc_pf_" + name)
    M, FL, per_call = [int(v) for v in g["params"]]
    x = g["x"].astype(np.float32) / 32768.0 * np.float32(g["scale"])
    sg = O.OracleSubbandGSC(_mic(M, 512, r=float(g["r"])), FL, (197, 0))
    outs = [sg.process(x[:, a:a + FL * per_call], postfilter=True)[0] for a in range(0, x.shape[1], FL * per_call)]
    out = np.concatenate(outs)
    assert rms(out - g["output"]) < 1e-6 * rms(g["output"])
    om = sg.omlsa_multi
    assert np.allclose(om.G, g["omlsa_G"], rtol=1e-6, atol=1e-8) and np.allclose(om.p, g["omlsa_p"], rtol=1e-6, atol=1e-9)
    # lambda_d stays at the first frame's power here (p == 1 from the second frame on: alpha_tilde == 1): the power of an output block that
    # is still all but zero, i.e. numbers at the transform's rounding floor (1e-22 .. 1e-14) — held at the bins that carry a value
    ref = g["omlsa_lambda_d"]
    rel = np.abs(om.lambda_d - ref) / ref
    assert np.median(rel) < 1e-4 and np.max(rel[ref > 1e-3 * ref.max()]) < 2e-2
    assert np.allclose(om.xi_hat, g["omlsa_xi_hat"], rtol=1e-4, atol=1e-9) and np.allclose(om.q_hat, g["omlsa_q_hat"], rtol=1e-5, atol=1e-6)
    # and the five results are those of postfilter=False
    sg0 = O.OracleSubbandGSC(_mic(M, 512, r=float(g["r"])), FL, (197, 0))
    out0 = np.concatenate([sg0.process(x[:, a:a + FL * per_call])[0] for a in range(0, x.shape[1], FL * per_call)])
    assert np.array_equal(out0, out)


def test_td_filters(golden):
    g = golden("g13_tdfilters")
    x, d = g["x"], g["d"]
    nl, rl = O.OracleNlms(64, 0.1), O.OracleRls(32)
    e1 = np.array([nl.update(x[i], d[i])[0] for i in range(x.size)])
    e2 = np.array([rl.update(x[i], d[i])[0] for i in range(x.size)])
    assert np.allclose(e1, g["e_nlms"], rtol=1e-9, atol=1e-12) and np.allclose(nl.w, g["w_nlms"], rtol=1e-9, atol=1e-12)
    assert np.allclose(e2, g["e_rls"], rtol=1e-6, atol=1e-9) and np.allclose(rl.w, g["w_rls"], rtol=1e-6, atol=1e-9)
    l2 = O.OracleNlms(300, 0.2, normalization=False)
    e3 = np.array([l2.update(x[i] * 0.1, d[i] * 0.1, p=0.5)[0] for i in range(1000)])
    assert np.allclose(e3, g["e_lms"], rtol=1e-9, atol=1e-12)


def _run_fdaf(f, x, d, p, trunc):
    hop = f.hop_len
    nb = x.shape[0] // hop
    e = np.zeros(nb * hop)
    for n in range(nb):
        pn = p[n] if np.ndim(p) == 1 else p[n][:, None]
        en, _ = f.update(x[n * hop:(n + 1) * hop], d[n * hop:(n + 1) * hop], p=pn, fir_truncate=trunc)
        e[n * hop:(n + 1) * hop] = en[:, 0]
    return e


@pytest.mark.parametrize("case,kind", [("a", "plain"), ("b", "plain"), ("c", "bm"), ("d", "aic")])
def test_fdaf(golden, case, kind):
    g = golden("g14_fdaf")
    L, C, mu, alpha, nc, trunc = g[case + "_params"]
    f = O.OracleFastFreqLms(filter_len=int(L), mu=float(mu), n_channels=int(C), alpha=float(alpha), non_causal=bool(nc),
                            kind=kind, weight_norm=(kind == "aic"))
    e = _run_fdaf(f, g[case + "_x"], g[case + "_d"], g[case + "_p"], None if trunc < 0 else int(trunc))
    assert np.allclose(e, g[case + "_e"], rtol=1e-9, atol=1e-12)
    assert np.allclose(f.w, g[case + "_w"], rtol=1e-9, atol=1e-12)
    assert np.allclose(f.W, g[case + "_W"], rtol=1e-9, atol=1e-12)
    assert np.allclose(f.P[:, 0], g[case + "_P"], rtol=1e-9, atol=1e-12)
    if case == "d":
        assert float(g["d_final_norm"]) > 0.0029          # the norm limiter was active in the fixture


@pytest.mark.parametrize("name", ["rec1", "rec1_pf", "synth_m6_pf"])
def test_tdgsc(golden, name):
    g = golden("g15_tdgsc_" + name)
    M, FL, pf = [int(v) for v in g["params"]]
    x = g["x"].astype(np.float32) / 32768.0 if g["x"].dtype == np.int16 else g["x"]
    o = O.OracleTDGSC(_mic(M, 512, r=float(g["r"])), frameLen=FL, angle_deg=(197, 0))
    out, p, obm = o.process(x.T.astype(np.float64), postfilter=bool(pf))
    assert np.allclose(p, g["p"], atol=1e-12)
    assert rms(obm - g["output_bm"]) < 1e-6 * rms(g["output_bm"])
    assert rms(out - g["output"]) < 1e-7 * rms(g["output"])
    assert np.allclose(o.aic_filter.w, g["w"], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("name", ["rec1", "rec1_pf", "synth_m6_pf", "burst", "burst64"])
def test_fdgsc(golden, name):
    g = golden("g16_fdgsc_" + name)
    M, FL, pf = [int(v) for v in g["params"]]
    x = g["x"].astype(np.float32) / 32768.0 if g["x"].dtype == np.int16 else g["x"]
    o = O.OracleFDGSC(_mic(M, 2 * FL, r=float(g["r"])), frameLen=FL, angle_deg=(197, 0))
    out, p, fix, fix_d, bm, al, al_d = o.process(x.T.astype(np.float64), postfilter=bool(pf))
    assert np.allclose(p, g["p"], atol=1e-12)
    assert rms(fix - g["fix_output"]) < 1e-9 * rms(g["fix_output"])
    assert rms(fix_d - g["fix_output_delayed"]) < 1e-9 * rms(g["fix_output_delayed"])
    assert rms(bm - g["bm_output"]) < 1e-6 * rms(g["bm_output"])
    assert rms(al_d - g["aligned_output_delayed"]) < 1e-6 * rms(g["aligned_output_delayed"])
    assert rms(out - g["output"]) < 1e-7 * rms(g["output"])
    assert np.allclose(o.bm[0].w, g["w_bm0"], rtol=1e-7, atol=1e-10)


@pytest.mark.parametrize("tag", ["e", "f"])
def test_fdaf_two_path(golden, tag):
    """FastFreqLms(two_path=True): foreground / background filters with the 3 dB transfer rule (FastFreqLms.py:94-104,162-176)."""
    g = golden("g14b_fdaf_two_path")
    L, C, mu, alpha = g[tag + "_params"]
    L, C = int(L), int(C)
    f = O.OracleFastFreqLms(filter_len=L, mu=float(mu), n_channels=C, alpha=float(alpha), two_path=True)
    x, d = g[tag + "_x"], g[tag + "_d"]
    e = np.zeros_like(d)
    transfers = []
    for n in range(d.size // L):
        fg0 = f.foreground.copy()
        en, w = f.update(x[n * L:(n + 1) * L], d[n * L:(n + 1) * L])
        e[n * L:(n + 1) * L] = en[:, 0]
        transfers.append(int(not np.array_equal(fg0, f.foreground)))
    assert np.array_equal(transfers, g[tag + "_transfers"])
    assert rms(e - g[tag + "_e"]) < 1e-9 * rms(g[tag + "_e"]) and rms(f.foreground - g[tag + "_F"]) < 1e-9 * rms(g[tag + "_F"])


def test_transform_custom_window(golden):
    """Transform(window=...) with a window of n_fft samples (transform.py:415-416)."""
    g = golden("g1b_transform_window")
    nfft, hop, M = [int(v) for v in g["params"]]
    t = O.OracleTransform(channel=M, n_fft=nfft, hop_length=hop, window=g["window"])
    Y = t.stft(g["x"])
    assert np.array_equal(Y.astype(np.complex64), g["Y"])
    assert np.allclose(t.istft(Y), g["y"], rtol=1e-7, atol=1e-9)


# ---- the reference's whole recording (26.7 s, 1 670 hops) and its real 8-channel recording ----------------------
def _last(a, hops, hop=256):
    return a[-hops * hop:]


def test_long_adaptive_mvdr(golden):
    g, x = golden("g17_adaptive_rec1_full"), golden("g17_rec1_full")["x"].astype(np.float32) / 32768.0
    ab = O.OracleAdaptiveMVDR(_mic(4, 512), frameLen=512, hop=256, nfft=512)
    T = x.shape[1] // 256
    ys = []
    for t in range(T):                                        # hop by hop, for the state snapshots
        ys.append(ab.process(x[:, t * 256:(t + 1) * 256], ANGLE, method=2))
        if t in (1, 500, 1000):
            assert np.allclose(ab.Rvv, g["Rvv_t%d" % t], rtol=2e-6, atol=1e-12), t          # fixture stores complex64
            assert np.allclose(ab.mcra.p, g["p_t%d" % t], rtol=1e-6, atol=1e-9), t
    y = np.concatenate(ys)
    assert rms(y - g["y"]) < 2e-7 * rms(g["y"]) and rms(_last(y - g["y"], 200)) < 2e-7 * rms(_last(g["y"], 200))   # float32 storage
    assert np.allclose(ab.Rvv, g["Rvv"], rtol=1e-9, atol=1e-14) and np.allclose(ab.mcra.p, g["mcra_p"], rtol=1e-10, atol=1e-14)


def test_long_gsc(golden):
    g, x = golden("g17_gsc_rec1_full"), golden("g17_rec1_full")["x"].astype(np.float32) / 32768.0
    gsc = O.OracleGSC(_mic(4, 512), frameLen=512)
    y = gsc.process(x, ANGLE, method=2)
    assert rms(y - g["y"]) < 2e-7 * rms(g["y"]) and rms(_last(y - g["y"], 200)) < 2e-7 * rms(_last(g["y"], 200))
    assert np.allclose(gsc.G, g["G"], rtol=1e-7, atol=1e-10) and np.allclose(gsc.spp.p, g["spp_p"], rtol=1e-6, atol=1e-9)


def test_long_subband_gsc(golden):
    g, x = golden("g17_subbandgsc_rec1_full"), golden("g17_rec1_full")["x"].astype(np.float32) / 32768.0
    sg = O.OracleSubbandGSC(_mic(4, 512), 256, (197, 0))
    out, fix, bm, p, al = sg.process(x)
    assert rms(out - g["output"]) < 1e-6 * rms(g["output"]) and rms(_last(out - g["output"], 200)) < 1e-6 * rms(_last(g["output"], 200))
    assert rms(fix - g["fix_output"]) < 1e-6 * rms(g["fix_output"]) and rms(bm[:, ::4] - g["bm_output"]) < 1e-6 * rms(g["bm_output"])
    assert np.max(np.abs(p - g["p"])) < 1e-5


def test_an101_adaptive_mvdr_8_channels(golden):
    g = golden("g18_adaptive_an101")
    M, nfft, hop, method = [int(v) for v in g["params"]]
    x = g["x"].astype(np.float32) / 32768.0
    ab = O.OracleAdaptiveMVDR(_mic(M, nfft, r=float(g["r"]), atype="linear"), frameLen=nfft, hop=hop, nfft=nfft)
    y = ab.process(x, ANGLE, method=method)
    assert rms(y - g["y"]) < 1e-7 * rms(g["y"])
    assert np.allclose(ab.Rvv, g["Rvv"], rtol=1e-9, atol=1e-14) and np.allclose(ab.H, g["H"], rtol=1e-5, atol=1e-7)


def test_an101_wpe_8_channels(golden):
    """BASELINE config 4's WPE shape (8 channels x 2 taps, 1024/512) on the real recording, against the *patched* reference (R6, R7)."""
    g, x = golden("g18_wpe_an101"), golden("g18_adaptive_an101")["x"].astype(np.float32) / 32768.0
    C, N, D, nb, hop = [int(v) for v in g["params"]]
    wpe = O.OracleWpe(channels=C, filter_len=N, num_bands=nb, delay=D, hop_length=hop)
    xt = x.T
    y = np.concatenate([wpe.update(xt[n:n + hop])[0] for n in range(0, xt.shape[0], hop)])
    assert rms(y - g["y"]) < 1e-7 * rms(g["y"])
    assert np.allclose(wpe.W, g["W"], rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("name", ["nb_c4n20", "c8n10"])
def test_wpe_wide_taps(golden, name):
    """the reference's maintained operating point Wpe(channels=4, filter_len=20, delay=4, num_bands=256, hop_length=64) (example/wpe.ipynb
    cell 2) on rec1, and 8 channels x 10 taps at 1024 / 512 (SURVEY 8d's cfg4 sizing) on an101: against the *patched* reference (R6, R7)."""
    g = golden("g21_wpe_" + name)
    C, N, D, nb, hop = [int(v) for v in g["params"]]
    x = (g["x"].astype(np.float32) / 32768.0).T
    wpe = O.OracleWpe(channels=C, filter_len=N, num_bands=nb, delay=D, hop_length=hop)
    T = x.shape[0] // hop
    ys = []
    for n in range(T):
        ys.append(wpe.update(x[n * hop:(n + 1) * hop])[0])
        if n == T // 2 - 1:
            assert np.allclose(wpe.W[::8], g["W_mid"], rtol=1e-4, atol=1e-7)
    y = np.concatenate(ys)
    assert rms(y - g["y"]) < 1e-6 * rms(g["y"])                       # (the fixture stores y as float32)
    assert np.allclose(wpe.W[g["bins"]], g["W"], rtol=1e-4, atol=1e-7)
    assert np.allclose(wpe.P[g["bins_P"]], g["P"], rtol=1e-4, atol=1e-9)


def test_gev_flow_and_pmwf_weight(golden):
    """mvdr.ipynb's GEV flow (get_gev_vector -> phase_correction -> blind_analytic_normalization) and the free compute_pmwf_weight
    (beamformer/beamformer.py:34-130; R10) — scipy's eigh is the same third-party routine the fixture was made with."""
    g = golden("g19_gev")
    W = O.get_gev_vector(g["Phi_xx"], g["Phi_vv"])
    assert np.allclose(W, g["W_gev"], rtol=1e-9, atol=1e-12)
    Wp = O.phase_correction(W)
    assert np.allclose(Wp, g["W_pc"], rtol=1e-9, atol=1e-12)
    Wb = O.blind_analytic_normalization(Wp, g["Phi_vv"])
    assert np.allclose(Wb, g["W_ban"], rtol=1e-9, atol=1e-12)
    x = g["x"].astype(np.float32) / 32768.0
    tr = O.OracleTransform(channel=4, n_fft=512, hop_length=256)
    D = tr.stft(x.T)
    y = tr.istft(np.einsum("inj,ij->in", D, Wb.conj())[:, :, None])
    assert rms(np.asarray(y).reshape(-1) - np.asarray(g["y"]).reshape(-1)) < 1e-7 * rms(g["y"])
    for beta in (1, 10):
        assert np.allclose(O.compute_pmwf_weight(g["xi"], g["Phi_xx"], np.linalg.inv(g["Phi_vv"]), beta), g["w_pmwf_b%d" % beta], rtol=1e-9, atol=1e-12)
