"""CPU oracle for the DistantSpeech per-frame enhancement hot path (NumPy restatement).

TEST INFRASTRUCTURE — NOT PRODUCT CODE.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the *checker* (or the timed CPU baseline), never as the thing shipped.
The product path (``distantspeech_amd``) never imports it and fails loudly without its HIP library.

What it is: a vectorised-over-bins NumPy restatement of the reference algorithms, each function
citing the reference ``file:line`` (paths relative to /root/reference) it follows.  Arithmetic is
float64/complex128 like the reference (``dtype=np.float32`` gives an all-fp32 restatement used to
predict the fp32 GPU kernels' error).

Parity pin: the reference holds no golden vectors or known-answer tests for this path
(SURVEY.md §4), so this oracle is pinned against outputs of the reference itself, generated in
the build container by ``tests/golden/make_golden.py`` (which imports the reference from
/root/reference through ``tests/golden/_ref_shim.py``) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against those fixtures.

Third-party arithmetic the reference calls and which is not under /root/reference
(numpy.fft.rfft/irfft, numpy.linalg.inv, scipy.signal.get_window/convolve; reference pins
numpy==1.21.6, scipy==1.8.0, librosa==0.9.1 in requirements.txt:57,74,110) is used here at the
versions of this image; the fixtures pin the results at those versions.
"""
import numpy as np

__all__ = [
    "sqrt_hann", "OracleTransform", "OracleMicArray", "compute_tau", "gen_noise_msc",
    "steering_from_doa", "fixed_weights", "circular_tao", "OracleMCRA", "OracleAdaptiveMVDR",
    "OracleFixedBeamformer", "OracleMcMcra", "OracleMcSppBase", "OracleMcCDR", "OracleMcSpp", "steering",
    "compute_mvdr_weight", "OracleOmlsaMulti", "OracleGSC",
    "OracleSubbandLMS", "OracleSubbandLmsMc", "OracleSubbandRLS", "OracleWpe", "fractional_delay_filter_bank",
    "OracleNlms", "OracleRls", "OracleDcNotch", "OracleTimeAlignment", "OracleDelaySamples", "OracleSubbandGSC", "OracleFastFreqLms", "OracleTDGSC", "OracleFDGSC", "OracleWpeMvdrPostfilter", "OracleMvdrPostfilter",
    "synth_utterance",
]


# --------------------------------------------------------------------------------------------
# transform/transform.py
# --------------------------------------------------------------------------------------------
def sqrt_hann(n_fft):
    """sqrt of the periodic Hann window — transform/transform.py:418-419
    (librosa.filters.get_window('hann', n, fftbins=True) == scipy.signal.get_window, periodic:
    w[n] = 0.5 - 0.5 cos(2 pi n / N))."""
    n = np.arange(n_fft)
    return np.sqrt(0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft))


class OracleTransform:
    """Streaming STFT/ISTFT with carried overlap — transform/transform.py:407-481."""

    def __init__(self, channel=1, n_fft=256, hop_length=128, dtype=np.float64, window=None):
        self.channel = channel
        self.n_fft = n_fft
        self.hop_length = hop_length
        self.window = sqrt_hann(n_fft) if window is None else np.asarray(window, dtype=np.float64)   # :415-419
        self.half_bin = int(n_fft / 2 + 1)                  # :420
        self.overlap = n_fft - hop_length                   # :424
        self.previous_input = np.zeros((self.overlap, channel))   # :425
        self.previous_output = np.zeros((self.overlap, channel))  # :426
        self.W0 = np.sum(self.window ** 2)                  # :428
        self.rt = np.dtype(dtype)

    def stft(self, x):
        """x [samples, channels] (or [samples]) -> Y [half_bin, frames, channels] complex128 whose
        values are rounded to complex64 (module stft dtype default, transform.py:17,212,220)."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x[:, None]
        x = np.vstack((self.previous_input, x))             # :438
        n_frames = 1 + int((x.shape[0] - self.n_fft) / self.hop_length)   # :441
        idx = np.arange(self.n_fft)[:, None] + self.hop_length * np.arange(n_frames)[None, :]
        Y = np.zeros((self.half_bin, n_frames, self.channel), dtype=np.complex128)
        for ch in range(self.channel):
            frames = x[:, ch][idx]                          # util.frame, center=False (:209)
            if self.rt == np.float32:
                spec = np.fft.rfft((self.window.astype(np.float32)[:, None] * frames.astype(np.float32)), axis=0)
            else:
                spec = np.fft.rfft(self.window[:, None] * frames, axis=0)   # :220
            Y[:, :, ch] = spec.astype(np.complex64)         # :212 complex64 storage
            self.previous_input[:, ch] = x[-self.overlap:, ch]              # :451
        return Y

    def istft(self, Y):
        """Y [half_bin, frames, channels] -> [frames*hop] (or [frames*hop, channels]).
        irfft in float64, overlap-add into a float32 buffer (transform.py:243,359,234),
        carried tail (:476-477), scale hop/W0 (:479)."""
        Y = np.asarray(Y)
        if Y.ndim == 1:
            Y = Y[:, None, None]                            # :461-462
        if Y.ndim == 2:
            Y = Y[:, None, :]                               # :463-464 (2-D means [K, channels])
        half_bin, n_frames, n_channels = Y.shape
        assert n_channels <= self.channel
        hop, n_fft = self.hop_length, self.n_fft
        out = np.zeros((hop * n_frames, n_channels))
        for ch in range(n_channels):
            ytmp = self.window[:, None] * np.fft.irfft(Y[:, :, ch], axis=0)   # :368
            y = np.zeros(n_fft + hop * (n_frames - 1), dtype=np.float32)       # :358-359
            for f in range(n_frames):
                y[f * hop:f * hop + n_fft] += ytmp[:, f]                        # :232-234 (rounds to f32)
            y[: self.overlap] += self.previous_output[:, ch]                   # :476
            self.previous_output[:, ch] = y[-self.overlap:]                    # :477
            out[:, ch] = y[: -self.overlap] * hop / self.W0                    # :479
        return out.squeeze()


# --------------------------------------------------------------------------------------------
# beamformer/MicArray.py, gen_noise_msc.py, beamformer.py (geometry + fixed weights; host set-up)
# --------------------------------------------------------------------------------------------
class OracleMicArray:
    """Geometry subset of MicArray — beamformer/MicArray.py:20-75."""

    def __init__(self, arrayType="circular", r=0.032, c=343, M=4, n_fft=256, fs=16000):
        self.arrayType = arrayType
        self.r, self.c, self.M, self.n_fft, self.fs = r, c, M, n_fft, fs
        self.half_bin = round(n_fft / 2 + 1)
        self.gamma = np.arange(0, 360, int(360 / M)) * np.pi / 180        # :33
        self.omega = 2 * np.pi * np.arange(self.half_bin) * fs / n_fft     # :36
        self.mic_loc = np.zeros((M, 3))
        if arrayType == "circular":                                       # :62-66
            az = np.arange(0, 360, int(360 / M)) * np.pi / 180
            self.mic_loc[:, 0] = r * np.cos(0.0) * np.cos(az)
            self.mic_loc[:, 1] = r * np.cos(0.0) * np.sin(az)
            self.mic_loc[:, 2] = r * np.sin(0.0)
        elif arrayType == "linear":                                       # :67-68
            self.mic_loc[:, 0] = -(np.arange(M) - (M - 1) / 2) * r
        else:
            raise ValueError(arrayType)


def compute_tau(mic, incident_angle_rad):
    """Far-field delay per mic w.r.t. the origin — module fn beamformer/MicArray.py:149-187."""
    az, el = float(incident_angle_rad[0]), float(incident_angle_rad[1])
    p0 = -1 * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    tau = np.zeros((mic.M, 1))
    for m in range(mic.M):
        loc = -1 * mic.mic_loc[m, :]
        nrm = np.linalg.norm(loc)
        cos_theta = np.sum(loc * p0) / (np.linalg.norm(p0) * nrm + 1e-12)
        tau[m] = -1 * nrm * cos_theta / mic.c
    return tau


def gen_noise_msc(mic, nfft=256, Fvv_max=0.9998):
    """Diffuse-field coherence (sinc model) — beamformer/gen_noise_msc.py:7-28."""
    M, c, fs = mic.M, mic.c, mic.fs
    half_bin = round(nfft / 2 + 1)
    Fvv = np.zeros((half_bin, M, M))
    f = np.linspace(0, fs / 2, half_bin)
    f[0] = 1e-6
    for i in range(M):
        for j in range(M):
            dij = np.sqrt(np.sum((mic.mic_loc[i, :] - mic.mic_loc[j, :]) ** 2))
            if i == j:
                Fvv[:, i, j] = Fvv_max
            else:
                Fvv[:, i, j] = np.sin(2 * np.pi * f * dij / c) / (2 * np.pi * f * dij / c)
    return Fvv


def steering_from_doa(mic, nfft, look_angle_deg):
    """a0[k,m] = exp(-j w_k tau_m) — beamformer/beamformer.py:267-289 (with MicArray.n_fft == nfft,
    SURVEY §8a-5; bins >= mic.half_bin stay zero exactly like the reference loop :286)."""
    half_bin = round(nfft / 2 + 1)
    omega = 2 * np.pi * np.arange(half_bin) * mic.fs / nfft             # beamformer.py:248
    tau0 = compute_tau(mic, np.array(look_angle_deg) / 180 * np.pi)
    a0 = np.zeros((half_bin, mic.M), dtype=complex)
    kmax = min(half_bin, mic.half_bin)
    a0[:kmax, :] = np.exp(-1j * omega[:kmax, None] * tau0[None, :, 0])
    return a0


def fixed_weights(mic, nfft, look_angle_deg, weightType="DS", diag_value=1e-3):
    """DS: a0/M; SD: G^-1 a /(a^H G^-1 a), G = Fvv + diag — beamformer/beamformer.py:338-373,
    compute_mvdr_weight :133-155 (base-class semantics: Fvv = gen_noise_msc(mic, nfft), :262)."""
    a0 = steering_from_doa(mic, nfft, look_angle_deg)
    if weightType == "DS":
        return a0 / mic.M
    if weightType == "SD":
        Fvv = gen_noise_msc(mic, nfft)
        Ginv = np.linalg.inv(Fvv + np.eye(mic.M) * diag_value)
        num = Ginv @ a0[..., None]
        w = num / (a0[:, None, :].conj() @ num)
        return w[..., 0]
    raise ValueError(weightType)


def circular_tao(r, c, gamma, angle_rad):
    """tao_m = -r cos(el) cos(az - gamma_m)/c — adaptivebeamformer.py:52, GSC.py:186."""
    return -1 * r * np.cos(angle_rad[1]) * np.cos(angle_rad[0] - gamma) / c


# --------------------------------------------------------------------------------------------
# noise_estimation/NoiseEstimationBase.py + mcra.py
# --------------------------------------------------------------------------------------------
class OracleMCRA:
    """NoiseEstimationMCRA — noise_estimation/mcra.py:20-77, NoiseEstimationBase.py:5-60."""

    def __init__(self, nfft=256, p_max=0.999, p_min=1e-3, L=15, dtype=np.float64):
        rt = np.dtype(dtype).type
        self.rt = rt
        self.half_bin = int(nfft / 2 + 1)
        K = self.half_bin
        self.lambda_d = np.zeros(K, dtype=rt)
        self.alpha_d, self.alpha_s, self.delta_s, self.alpha_p = rt(0.95), rt(0.8), rt(5), rt(0.2)
        self.ell = 1                                        # Base :18
        self.b = [rt(0.25), rt(0.5), rt(0.25)]              # Base :19
        self.S = np.zeros(K, dtype=rt)
        self.Smin = np.zeros(K, dtype=rt)
        self.Stmp = np.zeros(K, dtype=rt)
        self.p = np.zeros(K, dtype=rt)
        self.alpha_tilde = np.zeros(K, dtype=rt)
        self.p_max, self.p_min = rt(p_max), rt(p_min)
        self.L = L                                          # mcra.py:25
        self.frm_cnt = 0

    def estimation(self, Y):
        rt = self.rt
        Y = np.asarray(Y)
        if np.iscomplexobj(Y):
            Y = np.abs(Y) ** 2                              # :29-30
        if Y.ndim > 1:
            Y = Y[:, 0]
        Y = Y.astype(rt)
        K = self.half_bin
        assert len(Y) == K
        one = rt(1)
        if self.frm_cnt == 0:                               # :38-41
            self.Smin[:K - 1] = Y[:K - 1]
            self.Stmp[:K - 1] = Y[:K - 1]
            self.lambda_d[:K - 1] = Y[:K - 1]
            self.p[:K - 1] = 0                              # :68-69 (frm_cnt < 2L)
        else:
            self.p[0] = 0                                   # :43-45
            sl = slice(1, K - 1)
            Sf = Y[0:K - 2] * self.b[0] + Y[1:K - 1] * self.b[1] + Y[2:K] * self.b[2]   # :46
            self.S[sl] = self.alpha_s * self.S[sl] + (one - self.alpha_s) * Sf          # :47
            self.Smin[sl] = np.minimum(self.Smin[sl], self.S[sl])                       # :49
            self.Stmp[sl] = np.minimum(self.Stmp[sl], self.S[sl])                       # :50
            if self.ell % self.L == 0:                      # :52-56 (applies to every bin of the frame)
                self.Smin[sl] = np.minimum(self.Stmp[sl], self.S[sl])
                self.Stmp[sl] = self.S[sl]
                self.ell = 0
            Sr = self.S[sl] / (self.Smin[sl] + rt(1e-6))    # :58
            I = (Sr > self.delta_s).astype(rt)              # :60-63
            self.p[sl] = self.alpha_p * self.p[sl] + (one - self.alpha_p) * I           # :65-67
            if self.frm_cnt < self.L * 2:                   # :68-69
                self.p[sl] = 0
        self.p = np.maximum(np.minimum(self.p, self.p_max), self.p_min)                 # :70
        self.frm_cnt += 1
        self.lambda_d[K - 1] = rt(1e-8)                     # :73
        self.ell += 1
        self.update_noise_psd(Y)
        return self.lambda_d

    def update_noise_psd(self, Y, beta=1.0):
        """NoiseEstimationBase.py:56-60."""
        rt = self.rt
        self.alpha_tilde = self.alpha_d + (rt(1) - self.alpha_d) * self.p
        self.lambda_d = self.alpha_tilde * self.lambda_d + rt(beta) * (rt(1) - self.alpha_tilde) * Y


def smooth_psd(x, previous_x, win, alpha):
    """3-tap frequency smoothing (zero-padded edges) + time smoothing —
    NoiseEstimationBase.py:33-51 (scipy.signal.convolve 'full', centre slice)."""
    xp = np.concatenate(([0.0], x, [0.0])).astype(x.dtype)
    sm = win[0] * xp[2:] + win[1] * xp[1:-1] + win[2] * xp[:-2]
    return alpha * previous_x + (1 - alpha) * sm


# --------------------------------------------------------------------------------------------
# beamformer/adaptivebeamformer.py (config 2) and fixedbeamformer.py (config 1)
# --------------------------------------------------------------------------------------------
METHODS = ["src", "DS", "MVDR", "TFGSC"]     # adaptivebeamformer.py:36


class OracleAdaptiveMVDR:
    """adaptivebeamfomer.process — beamformer/adaptivebeamformer.py:44-128: VAD-gated (estPos = None, :94) or with the noise covariance taken from
    the first `estPos` (frame, bin) slots after a restart (:90-93; the count restarts when the look direction or the method changes, :70-79).

    Defined for any number of hops per call as "T successive one-hop calls" (SURVEY §8b chunking
    contract; the reference is only valid one hop per call at HEAD, adaptivebeamformer.py:122)."""

    def __init__(self, mic, frameLen=512, hop=None, nfft=None, dtype=np.float64, mcra_L=15):
        self.M = mic.M
        self.nfft = int(nfft) if nfft else int(frameLen)
        self.hop = int(hop) if hop else int(frameLen // 2)
        self.half_bin = round(self.nfft / 2 + 1)
        self.r, self.c, self.fs, self.gamma = mic.r, mic.c, mic.fs, mic.gamma
        self.omega = 2 * np.pi * np.arange(self.half_bin) * self.fs / self.nfft   # :20
        self.rt = np.dtype(dtype).type
        self.ct = np.complex64 if self.rt == np.float32 else np.complex128
        K, M = self.half_bin, self.M
        self.H = np.ones((M, K), dtype=self.ct) / M                     # :22
        self.Rvv = np.zeros((K, M, M), dtype=self.ct)                   # :32-34
        self.Rvv_inv = np.zeros((K, M, M), dtype=self.ct)
        self.Ryy = np.zeros((K, M, M), dtype=self.ct)
        self.transformer = OracleTransform(n_fft=self.nfft, hop_length=self.hop, channel=M, dtype=dtype)
        self.mcra = OracleMCRA(nfft=self.nfft, L=mcra_L, dtype=dtype)   # :40
        self.estPos = None                                              # :30
        self.frameCount = 0                                             # :28
        self.angle = np.array([0, 0]) / 180 * np.pi                     # :24
        self.AlgorithmIndex = 0                                         # :37

    def process_frame(self, Zk, angle_rad, method=2):
        """Zk [K, M] one STFT frame -> Y [K].  :69-120."""
        rt, ct = self.rt, self.ct
        K, M = self.half_bin, self.M
        tao = circular_tao(self.r, self.c, self.gamma, angle_rad)       # :52
        a = np.exp(-1j * self.omega[:, None] * tao[None, :]).astype(ct)  # :84  [K, M]
        Z = Zk.astype(ct)
        alpha_y, alpha_v, diag = rt(0.8), rt(0.9998), rt(1e-6)          # :65-66,89
        self.mcra.estimation(np.abs(Z[:, 0] * np.conj(Z[:, 0])))        # :81
        zz = Z[:, :, None] * np.conj(Z[:, None, :])                     # z z^H  [K, M, M]
        self.Ryy = alpha_y * self.Ryy + (rt(1) - alpha_y) * zz          # :86-88
        if not np.array_equal(np.asarray(angle_rad), self.angle) or method != self.AlgorithmIndex:   # :70-79
            self.angle = np.array(angle_rad, dtype=float)
            self.AlgorithmIndex = method
            self.frameCount = 0
        if self.estPos is not None:                                     # :90-93: frameCount advances once per BIN while below estPos
            n = min(max(int(self.estPos) - self.frameCount, 0), K)
            upd = np.arange(K) < n
            self.frameCount += n
        else:
            upd = self.mcra.p < rt(0.4)                                 # :94
        self.Rvv[upd] = alpha_v * self.Rvv[upd] + (rt(1) - alpha_v) * zz[upd]   # :97-99
        if upd.any():
            self.Rvv_inv[upd] = np.linalg.inv(self.Rvv[upd] + diag * np.eye(M, dtype=rt))   # :103-104
        name = METHODS[method]
        if name == "src":                                               # beamformer.py:320-322
            H = np.zeros((K, M), dtype=ct)
            H[:, 0] = a[:, 0]
        elif name == "DS":                                              # beamformer.py:323-324
            H = a / M
        elif name == "MVDR":                                            # beamformer.py:325-326
            num = (self.Rvv_inv @ a[:, :, None])[..., 0]
            den = np.sum(np.conj(a) * num, axis=1)
            H = num / den[:, None]
        elif name == "TFGSC":                                           # beamformer.py:327-333
            temp = self.Rvv_inv @ self.Ryy
            tr = np.trace(temp, axis1=1, axis2=2)
            col0 = temp[:, :, 0].copy()
            col0[:, 0] -= 1
            H = col0 / (tr - M)[:, None]
        else:
            raise ValueError(name)
        self.H = H.T.copy()                                             # [M, K] like the reference
        return np.sum(np.conj(H) * Z, axis=1)                           # :119-120

    def beampattern(self, omega, H):
        """beamformer.beampattern, beamformer.py:536-553: [360, half_bin] in dB; r = 0.032 hard-wired there."""
        half_bin = H.shape[1]
        out = np.zeros((360, half_bin))
        for az in range(360):
            tao = -1 * 0.032 * np.cos(0) * np.cos(az * np.pi / 180 - self.gamma) / self.c
            a = np.exp(-1j * omega[None, :] * tao[:, None])              # [M, K]
            out[az] = np.abs(np.sum(np.conj(H) * a, axis=0))
        with np.errstate(divide="ignore"):
            return 10 * np.log10(out)

    def process(self, x, angle_rad, method=2):
        """x [M, T*hop] -> y [T*hop]; equals T one-hop reference calls concatenated."""
        X = self.transformer.stft(np.asarray(x).T)                      # :49
        T = X.shape[1]
        out = []
        for t in range(T):   # one ISTFT call per hop, exactly like T one-hop reference calls (:122)
            Yt = self.process_frame(X[:, t, :], angle_rad, method)
            out.append(np.atleast_1d(self.transformer.istft(Yt[:, None, None])))
        return np.concatenate(out)


class OracleFixedBeamformer:
    """FixedBeamformer.process (DS or SD weights) — beamformer/fixedbeamformer.py:147-207 with the
    base-class weight semantics (beamformer.py:338-373) because the subclass's gen_noise_msc call
    has a shape bug at nfft != 256 (fixedbeamformer.py:140)."""

    def __init__(self, mic, frameLen=512, hop=None, nfft=None, angle=(197, 0), weightType="DS", dtype=np.float64):
        self.mic = mic
        self.M = mic.M
        self.nfft = int(nfft) if nfft else int(frameLen)
        self.hop = int(hop) if hop else int(frameLen // 2)
        self.half_bin = round(self.nfft / 2 + 1)
        self.weightType = weightType
        self.angle = list(angle)
        self.W = fixed_weights(mic, self.nfft, self.angle, weightType)
        self.transform = OracleTransform(n_fft=self.nfft, hop_length=self.hop, channel=self.M, dtype=dtype)

    def process(self, x, angle=None):
        """x [samples, channels] -> [samples]."""
        if angle is not None and list(angle) != self.angle:
            self.angle = list(angle)
            self.W = fixed_weights(self.mic, self.nfft, self.angle, self.weightType)
        D = self.transform.stft(x)
        Yf = np.einsum("kc,ktc->kt", self.W.conj(), D)[:, :, None]     # :163 per frame
        return np.atleast_1d(self.transform.istft(Yf))


# --------------------------------------------------------------------------------------------
# noise_estimation/mc_mcra.py
# --------------------------------------------------------------------------------------------
class OracleMcMcra:
    """McMcra (multichannel SPP with real covariances + Wiener-type gain) — mc_mcra.py:25-224."""

    def __init__(self, nfft=256, channels=4, dtype=np.float64):
        rt = np.dtype(dtype).type
        self.rt = rt
        self.M = channels
        self.half_bin = int(nfft / 2 + 1)
        K, M = self.half_bin, channels
        self.alpha_d, self.alpha = rt(0.95), rt(0.92)                   # :36,38
        self.psi_0, self.psi_tilde_0 = rt(100), rt(100)                 # :60-61
        self.p = np.zeros(K, dtype=rt)
        self.G = np.zeros(K, dtype=rt)
        self.G_H1 = np.zeros(K, dtype=rt)
        self.q_local = np.ones(K, dtype=rt) * rt(0.999)
        self.Phi_yy = np.zeros((K, M, M), dtype=rt)                     # reference layout [M, M, K] (:68-69)
        self.Phi_vv = np.zeros((K, M, M), dtype=rt)
        self.xi = np.zeros(K, dtype=rt)
        self.gamma = np.zeros(K, dtype=rt)
        self.psi = np.zeros(K, dtype=rt)
        self.psi_tilde = np.zeros(K, dtype=rt)
        self.frm_cnt = 0

    def estimation(self, y):
        """y [K, M] complex — :179-208."""
        rt = self.rt
        M = self.M
        y = np.asarray(y)
        yy = np.real(np.conj(y[:, :, None]) * y[:, None, :]).astype(rt)     # Re(conj(y)^T y)  :182-184
        self.Phi_yy = self.alpha * self.Phi_yy + (rt(1) - self.alpha) * yy
        if self.frm_cnt < 5:                                                # :186-187
            self.Phi_vv = self.Phi_yy.copy()
        Phi_xx = self.Phi_yy - self.Phi_vv                                  # :189
        inv = np.linalg.inv(self.Phi_vv + np.eye(M, dtype=rt) * rt(1e-6))   # :191
        tr = np.trace(inv @ self.Phi_yy, axis1=1, axis2=2)
        self.xi = np.minimum(np.maximum(tr - M, rt(1e-6)), rt(1e6))         # :193-194
        A = inv @ Phi_xx @ inv
        # Re(conj(y) A y^T) with real A: sum_ij A_ij Re(conj(y_i) y_j)       :196-198
        g = np.real(np.einsum("ki,kij,kj->k", np.conj(y), A.astype(np.complex128 if rt == np.float64 else np.complex64), y))
        self.gamma = np.minimum(np.maximum(g.astype(rt), rt(1e-6)), rt(1e6))            # :199
        # compute_q_local :91-105
        self.psi = np.real(np.einsum("ki,kij,kj->k", y, inv.astype(y.dtype), np.conj(y))).astype(rt)
        self.psi_tilde = tr.astype(rt)
        q_max, q_min = rt(0.99), rt(0.01)
        lin = (self.psi_tilde_0 - self.psi_tilde) / (self.psi_tilde_0 - M)
        lin = np.minimum(np.maximum(lin, q_min), q_max)
        q = np.where((self.psi >= self.psi_0) | (self.psi_tilde > self.psi_tilde_0), q_min,
                     np.where(self.psi_tilde < M, q_max, lin)).astype(rt)
        self.q_local = q
        self.q = q                                                          # :138
        # compute_p(p_max=0.99, p_min=0.01) :143-151,204
        p = 1 / (1 + q / (1 - q) * (1 + self.xi) * np.exp(-1 * (self.gamma / (1 + self.xi))))
        self.p = np.minimum(np.maximum(p, rt(0.01)), rt(0.99)).astype(rt)
        # update_noise_psd :210-224
        at = (self.alpha_d + (rt(1) - self.alpha_d) * self.p)[:, None, None]
        self.Phi_vv = (at * self.Phi_vv + (rt(1) - at) * yy).astype(rt)
        # compute_weight :153-157
        Gmin = rt(0.0631)
        self.G_H1 = self.xi / (1 + self.xi)
        G = np.power(self.G_H1, self.p) * np.power(Gmin, (1 - self.p))
        G = np.maximum(np.minimum(G, rt(1)), Gmin)
        G[:2] = 0
        self.G = G.astype(rt)
        self.frm_cnt += 1


class OracleMcSppBase:
    """McSppBase.estimation + compute_pmwf_weight — noise_estimation/mcspp_base.py:28-324."""

    def __init__(self, nfft=256, channels=4):
        self.M = channels
        self.half_bin = int(nfft / 2 + 1)
        K, M = self.half_bin, channels
        self.alpha_d, self.alpha = 0.92, 0.92                                # :37,39
        self.p = np.zeros(K)
        self.q = np.ones(K) * 0.6
        self.w = np.zeros((K, M), dtype=complex)
        self.Phi_yy = np.zeros((K, M, M), dtype=complex)
        self.Phi_vv = np.zeros((K, M, M), dtype=complex)
        self.Phi_vv_inv = np.zeros((K, M, M), dtype=complex)
        self.Phi_xx = np.zeros((K, M, M), dtype=complex)
        self.xi = np.zeros(K)
        self.gamma = np.zeros(K)
        self.mcra = OracleMCRA(nfft=nfft, L=15)                              # :75-76
        self.frm_cnt = 0

    def estimation(self, y):
        M = self.M
        y = np.asarray(y, dtype=complex)
        psd_yy = np.einsum('ij,il->ijl', y, y.conj())                        # :88
        self.Phi_yy = self.alpha * self.Phi_yy + (1 - self.alpha) * psd_yy   # :90
        self.Phi_xx = self.Phi_yy - self.Phi_vv                              # :275
        inv = np.linalg.inv(self.Phi_vv.real + np.eye(M) * 1e-6)             # :277-279
        self.Phi_vv_inv = inv.astype(complex)
        xi = np.trace(inv @ self.Phi_xx.real, axis1=-2, axis2=-1)            # :281
        g = (y[:, None, :].conj() @ inv @ self.Phi_xx.real @ inv @ y[:, :, None]).real.squeeze()   # :283-285
        self.xi = np.minimum(np.maximum(xi, 1e-6), 1e6)
        self.gamma = np.minimum(np.maximum(g, 1e-6), 1e6)
        self.mcra.estimation(np.abs(y[:, 0] * np.conj(y[:, 0])))             # compute_q :113-118
        self.q = np.minimum(np.maximum(np.sqrt(1 - self.mcra.p), 0.01), 0.99)
        p = 1 / (1 + self.q / (1 - self.q) * (1 + self.xi) * np.exp(-1 * (self.gamma / (1 + self.xi))))   # :133
        self.p = np.minimum(np.maximum(p, 0.01), 0.99)
        at = (self.alpha_d + (1 - self.alpha_d) * self.p)[:, None, None]     # :314
        self.Phi_vv = at * self.Phi_vv + (1 - at) * psd_yy                   # :319-321
        self.w = (self.Phi_vv_inv @ self.Phi_xx)[:, :, 0] / (1 + self.xi[:, None])   # :238-240 (u = e_0, beta = 1)
        self.frm_cnt += 1
        return self.p


class OracleMcCDR:
    """McCDR.estimation (coherent-to-diffuse-ratio speech presence prior) — noise_estimation/mccdr.py:25-177,
    with the pieces of coherence/BinauralEnhancement.py it drives (update_CSD_PSD :33-62, updateMSC :24-31).
    Only the (1, 2) microphone pair enters the result (mccdr.py:141-143)."""

    def __init__(self, nfft=256, channels=4):
        self.M = channels
        self.half_bin = int(nfft / 2 + 1)
        K = self.half_bin
        mic = OracleMicArray(arrayType="circular", r=0.032, M=channels)            # mccdr.py:63
        self.Fn = gen_noise_msc(mic, nfft)[:, 1, 2]                                # BinauralEnhancement -> beamformer.Fvv
        self.Pxii = np.zeros((K, channels))
        self.Pxij12 = np.zeros(K, dtype=complex)
        self.mcra = OracleMCRA(nfft=nfft, L=65)                                    # mccdr.py:60-61

    def estimation(self, y):
        alpha = 0.9                                                                # mccdr.py:131
        self.Pxii = alpha * self.Pxii + (1 - alpha) * np.real(y * y.conj())        # BinauralEnhancement.py:49-52
        self.Pxij12 = alpha * self.Pxij12 + (1 - alpha) * (y[:, 1] * y[:, 2].conj())   # :55-61 (pair (1,2))
        with np.errstate(divide="ignore", invalid="ignore"):
            Fx = self.Pxij12 / np.sqrt(self.Pxii[:, 1] * self.Pxii[:, 2])          # updateMSC :28
            Fn = self.Fn
            Fn2 = Fn ** 2
            Fx2 = np.abs(Fx) ** 2
            Gamma = (Fn * Fx.real - Fx2 - np.sqrt(Fn2 * Fx.real ** 2 - Fn2 * Fx2 + Fn2 - 2 * Fn * Fx.real + Fx2)) / (
                np.minimum(Fx2 - 1, -1e-3))                                        # mccdr.py:137-145
            Gamma = Gamma ** 2
            Gamma[Gamma > 1] = 1
            Gamma[Gamma < 0] = 1e-3
        self.mcra.estimation(y[:, 0])                                              # :174 (complex -> |.|^2)
        return np.sqrt(Gamma * self.mcra.p)                                        # :175


def steering(XXs):
    """principal eigenvector of each Hermitian matrix, phase-normalised by the reference sensor —
    beamformer/beamformer.py:10-31."""
    vs = np.linalg.eigh(XXs)[1][:, :, -1]
    return vs / np.exp(1j * np.angle(vs[:, 0:1]))


def compute_mvdr_weight(steer_vector, Rvv_inv):
    """w = R^-1 a / (a^H R^-1 a) — beamformer/beamformer.py:133-155."""
    num = Rvv_inv @ steer_vector[..., None]
    return (num / (steer_vector[:, None, :].conj() @ num))[..., 0]


def compute_pmwf_weight(xi, Rxx, Rvv_inv, beta=1):
    """parameterised multichannel Wiener filter w = (Rvv_inv Rxx) u / (beta + xi), u = e_0 — beamformer/beamformer.py:100-130 with the
    matrices as [bins, M, M] (the batched `Rvv_inv @ Rxx @ u` of :128 needs that layout; the reference's `channels = Rxx.shape[0]`, :124,
    only works where M == bins — repaired to Rxx.shape[1], make_golden.py R10)."""
    half_bin, channels = xi.shape[0], Rxx.shape[1]
    u = np.zeros((half_bin, channels, 1))
    u[:, 0, 0] = 1                                                                  # :125-127
    return (Rvv_inv @ Rxx @ u).squeeze() / (beta + xi[:, None])                     # :128


def get_gev_vector(target_psd_matrix, noise_psd_matrix):
    """principal generalised eigenvector per bin, scipy.linalg.eigh(a, b)[1][:, -1] — beamformer/beamformer.py:79-97.  Third-party
    arithmetic (scipy.linalg.eigh -> LAPACK zhegvd): the eigenvector's phase is LAPACK's."""
    from scipy.linalg import eigh
    bins, sensors, _ = target_psd_matrix.shape
    out = np.empty((bins, sensors), dtype=complex)
    for f in range(bins):
        try:
            out[f, :] = eigh(target_psd_matrix[f], noise_psd_matrix[f])[1][:, -1]  # :91-92
        except np.linalg.LinAlgError:
            out[f, :] = np.ones((sensors,)) / np.trace(noise_psd_matrix[f]) * sensors   # :95
    return out


def blind_analytic_normalization(vector, noise_psd_matrix, eps=0):
    """vector * |sqrt(v^H N N v)| / (|v^H N v| + eps) — beamformer/beamformer.py:34-63."""
    nominator = np.einsum('...a,...ab,...bc,...c->...', vector.conj(), noise_psd_matrix, noise_psd_matrix, vector)   # :55
    nominator = np.abs(np.sqrt(nominator))                                          # :56
    denominator = np.abs(np.einsum('...a,...ab,...b->...', vector.conj(), noise_psd_matrix, vector))                # :58-59
    return vector * (nominator / (denominator + eps))[..., np.newaxis]              # :61-62


def phase_correction(vector):
    """bin f rotated by exp(-j angle(sum_m w[f, m] conj(w[f - 1, m]))), in bin order — beamformer/beamformer.py:66-76."""
    w = vector.copy()
    for f in range(1, w.shape[0]):
        w[f, :] *= np.exp(-1j * np.angle(np.sum(w[f, :] * w[f - 1, :].conj(), axis=-1, keepdims=True)))              # :75
    return w


class OracleMcSpp:
    """McSpp.estimation — noise_estimation/mcspp.py:46-305 (q from McCDR, complex covariances, PMWF beta = 10)."""

    def __init__(self, nfft=256, channels=4):
        self.M, self.nfft = channels, nfft
        self.half_bin = int(nfft / 2 + 1)
        K, M = self.half_bin, channels
        self.mccdr = OracleMcCDR(nfft=nfft, channels=channels)
        self.alpha_d, self.alpha = 0.92, 0.92                                      # :60-61
        self.Phi_yy = np.zeros((K, M, M), dtype=complex)
        self.Phi_vv = np.zeros((K, M, M), dtype=complex)
        self.Phi_vv_inv = np.zeros((K, M, M), dtype=complex)
        self.Phi_xx = np.zeros((K, M, M), dtype=complex)
        self.p = np.zeros(K)
        self.w = np.zeros((K, M), dtype=complex)
        self.frm_cnt = 0

    def _core(self, y, diag_bin):
        M = self.M
        self.Phi_vv = 0.5 * (self.Phi_vv + np.conj(self.Phi_vv.swapaxes(-1, -2)))  # :210
        self.Phi_xx = self.Phi_yy - self.Phi_vv                                    # :212
        self.Phi_vv_inv = np.linalg.inv(self.Phi_vv + diag_bin)                    # :214
        xi = np.trace(np.real(self.Phi_vv_inv @ self.Phi_yy), axis1=-2, axis2=-1) - M   # :217
        index = np.where(xi < 0)
        if self.frm_cnt < 5:                                                       # :223-226
            self.Phi_vv_inv[index] = np.linalg.inv(self.Phi_yy[index] + diag_bin[index])
        else:
            self.Phi_vv_inv[index] = np.linalg.inv(self.Phi_yy[index])
        xi = np.trace(np.real(self.Phi_vv_inv @ self.Phi_yy), axis1=-2, axis2=-1) - M   # :228
        self.xi = np.minimum(np.maximum(xi, 1e-6), 1e8)                            # :230
        g = (y[:, None, :].conj() @ self.Phi_vv_inv @ self.Phi_yy @ self.Phi_vv_inv @ y[:, :, None]
             - y[:, None, :].conj() @ self.Phi_vv_inv @ y[:, :, None]).real.squeeze()   # :232-235
        self.gamma = np.minimum(np.maximum(g, 1e-6), 1e8)
        p = 1 / (1 + self.q / (1 - self.q) * (1 + self.xi) * np.exp(-1 * (self.gamma / (1 + self.xi))))   # compute_p
        self.p = np.minimum(np.maximum(p, 0.0), 1.0)

    def estimation(self, y, repeat=False):
        M = self.M
        y = np.asarray(y, dtype=complex)
        self.q = 1 - self.mccdr.estimation(y)                                      # compute_q :113-116
        fmin, fmax = int(500 * self.nfft / 16000), int(2000 * self.nfft / 16000)   # :258-259
        q_avg = np.mean(self.q[fmin:fmax])
        dv = q_avg * 1e-1 + (1 - q_avg) * 1e-4                                     # :254-262
        psd_yy = np.einsum('ij,il->ijl', y, y.conj())
        self.Phi_yy = self.alpha * self.Phi_yy + (1 - self.alpha) * psd_yy         # :266
        if self.frm_cnt < 10:                                                      # :273-275
            self.Phi_vv = self.Phi_yy.copy()
            self.q = np.full(self.half_bin, 0.99)
        diag_bin = np.array(np.broadcast_to(np.eye(M) * dv, (self.half_bin, M, M)))    # :203-204 (eye * (eye * dv))
        self._core(y, diag_bin)
        at = (self.alpha_d + (1 - self.alpha_d) * self.p)[:, None, None]           # update_noise_psd
        self.Phi_vv = at * self.Phi_vv + (1 - at) * psd_yy
        if repeat:                                                                 # :280-282
            self._core(y, diag_bin)
        self.w = (self.Phi_vv_inv @ self.Phi_xx)[:, :, 0] / (10 + self.xi[:, None])    # compute_pmwf_weight beta=10 :283
        self.frm_cnt += 1
        return self.p


# --------------------------------------------------------------------------------------------
# noise_estimation/omlsa_multi.py
# --------------------------------------------------------------------------------------------
class OracleOmlsaMulti:
    """NsOmlsaMulti (TBRR-OMLSA, Wiener G_H1) — omlsa_multi.py:27-156."""

    def __init__(self, nfft=256, M=4, cal_weights=False, dtype=np.float64):
        self.half_bin = int(nfft / 2 + 1)
        K = self.half_bin
        self.M = M
        self.G_H1 = np.ones(K)
        self.G = np.ones(K)
        self.Gmin = np.power(10, (-12 / 10))                 # :35-36
        self.gamma = np.ones(K)
        self.zeta_Y = np.ones(K)
        self.zeta_U = np.zeros((M - 1, K))
        self.MU_Y = np.ones(K)
        self.MU_U = np.zeros((M - 1, K))
        self.q_hat = np.ones(K)
        self.q_min, self.q_max = 1e-6, 0.9999998             # :54-55
        self.alpha_d = 0.85                                  # :57
        self.xi_hat = np.ones(K)
        self.p = np.zeros(K)
        self.lambda_d = np.zeros(K)
        self.first_frame = 1
        self.noise_est_fixed = OracleMCRA(nfft=nfft)         # :65-67 (L=15 default of NoiseEstimationMCRA)
        self.noise_est_ref = [OracleMCRA(nfft=nfft) for _ in range(M - 1)]
        self.win = np.array([0.25, 0.5, 0.25])
        self.alpha_s = 0.8
        self.cal_weights = cal_weights
        self.Omega = np.ones(K)
        self.gamma_s = np.ones(K)

    def estimation(self, y, u):
        """y [K] power of beam output, u [K, M-1] powers of the references — :73-156."""
        y = np.asarray(y, dtype=np.float64)
        u = np.asarray(u, dtype=np.float64)
        assert len(y) == self.half_bin
        self.MU_Y = self.noise_est_fixed.estimation(y)                      # :83
        for ch in range(self.M - 1):
            self.MU_U[ch, :] = self.noise_est_ref[ch].estimation(u[:, ch])  # :84-85
        if self.first_frame == 1:                                           # :87-93
            self.first_frame = 0
            self.lambda_d = y.copy()
            self.zeta_Y = y.copy()
            self.zeta_U = u[:, : self.M - 1].T.copy()                       # :91-92 (the first M - 1 columns, whatever u carries)
            return None
        alpha = 0.921
        self.zeta_Y = smooth_psd(y, self.zeta_Y, self.win, self.alpha_s)    # :98
        for ch in range(self.M - 1):
            self.zeta_U[ch, :] = smooth_psd(u[:, ch], self.zeta_U[ch, :], self.win, self.alpha_s)
        eps = 0.01
        self.Omega = np.maximum(self.zeta_Y - self.MU_Y, 1e-6) / (
            np.maximum(np.max(self.zeta_U - self.MU_U, axis=0), eps * self.MU_Y) + 1e-6)   # :107-109
        self.Omega = np.minimum(np.maximum(self.Omega, 0.1), 100)           # :110-111
        Bmin = 1.66
        self.gamma_s = np.minimum(y / (self.MU_Y * Bmin + 1e-6), 100)       # :115
        gamma_high, gamma_low, Omega_high, Omega_low = 10.0, 1.0, 3.0, 0.3  # :117-120
        q = np.maximum((gamma_high - self.gamma_s) / (gamma_high - gamma_low),
                       (Omega_high - self.Omega) / (Omega_high - Omega_low))
        q = np.where((self.gamma_s < gamma_low) | (self.Omega < Omega_low), 1.0, q)   # :122-129
        self.q_hat = np.minimum(np.maximum(q, self.q_min), self.q_max)      # :130
        gamma_pre = self.gamma.copy()
        self.gamma = y / np.maximum(self.lambda_d, 1e-10)                   # :134
        self.xi_hat = alpha * np.power(self.G_H1, 2) * gamma_pre + (1 - alpha) * np.maximum(self.gamma - 1, 0)  # :137
        nu = self.gamma * self.xi_hat / (1 + self.xi_hat)                   # :140
        self.G_H1 = self.xi_hat / (1 + self.xi_hat)                         # :144
        self.p = 1 / (1 + self.q_hat / (1 - self.q_hat) * (1 + self.xi_hat) * np.exp(-1 * nu))   # :147
        at = self.alpha_d + (1 - self.alpha_d) * self.p                     # Base :57 (alpha_d = 0.85 here)
        self.lambda_d = at * self.lambda_d + 1.47 * (1 - at) * y            # :149
        if self.cal_weights:                                                # :152-154
            G = np.power(self.G_H1, self.p) * np.power(self.Gmin, (1 - self.p))
            self.G = np.maximum(np.minimum(G, 1), self.Gmin)
        return self.lambda_d


# --------------------------------------------------------------------------------------------
# beamformer/GSC.py (config 3)
# --------------------------------------------------------------------------------------------
class OracleGSC:
    """GSC.process (frequency-domain GSC + SPP-controlled LMS AIC + McMcra gain) — GSC.py:174-294."""

    def __init__(self, mic, frameLen=512, dtype=np.float64, with_dead_state=True):
        self.M = mic.M
        self.nfft = int(frameLen)                           # beamformer.py:239-240 via GSC.py:33
        self.hop = int(frameLen // 2)
        self.half_bin = round(self.nfft / 2 + 1)
        self.r, self.c, self.fs, self.gamma = mic.r, mic.c, mic.fs, mic.gamma
        self.omega = 2 * np.pi * np.arange(self.half_bin) * self.fs / self.nfft
        K, M = self.half_bin, self.M
        self.G = np.zeros((M - 1, K), dtype=complex)        # :72
        self.U = np.zeros((M - 1, K), dtype=complex)
        self.Yfbf = np.zeros(K, dtype=complex)
        self.transformer = OracleTransform(n_fft=self.nfft, hop_length=self.hop, channel=M, dtype=dtype)
        self.spp = OracleMcMcra(nfft=self.nfft, channels=M)                 # :80-81
        self.with_dead_state = with_dead_state
        self.mcra = OracleMCRA(nfft=self.nfft)                              # :77 (output-dead)
        self.omlsa_multi = OracleOmlsaMulti(nfft=self.nfft, cal_weights=True, M=M)   # :78 (output-dead)

    def process_frame(self, Zk, angle_rad, method=2):
        M = self.M
        mu = 0.01                                                            # :202
        tao = circular_tao(self.r, self.c, self.gamma, angle_rad)           # :186,209
        a = np.exp(-1j * self.omega[:, None] * tao[None, :])                # [K, M]
        self.spp.estimation(Zk)                                              # :225
        if self.with_dead_state:
            self.mcra.estimation(np.abs(Zk[:, 0] * np.conj(Zk[:, 0])))      # :240
        if method == 0:
            return Zk[:, 0].copy()                                           # :242-243
        W = a / np.sum(np.conj(a) * a, axis=1, keepdims=True)               # :219  W = a/(a^H a)
        # U_i = BM[:, i]^H z = conj(a_0) z_0 - conj(a_{i+1}) z_{i+1}          :220-222,261
        U = np.conj(a[:, 0:1]) * Zk[:, 0:1] - np.conj(a[:, 1:]) * Zk[:, 1:]  # [K, M-1]
        Yfbf = np.sum(np.conj(W) * Zk, axis=1)                               # :263
        G = self.G.T                                                         # [K, M-1]
        Y = Yfbf - np.sum(np.conj(G) * U, axis=1)                            # :266
        G = G + mu * (1 - self.spp.p)[:, None] * U * np.conj(Y)[:, None]     # :270-274 (Pest == 1)
        self.G = G.T.copy()
        self.U = U.T.copy()
        self.Yfbf = Yfbf
        if self.with_dead_state:
            self.omlsa_multi.estimation(np.real(Y * np.conj(Y)), np.real(U * np.conj(U)))   # :281-283
        return Y * self.spp.G                                                # :286

    def process(self, x, angle_rad, method=2):
        """x [M, T*hop] -> y [T*hop] (T successive one-hop reference calls)."""
        X = self.transformer.stft(np.asarray(x).T)
        T = X.shape[1]
        out = []
        for t in range(T):   # one ISTFT call per hop, exactly like T one-hop reference calls (:288)
            Yt = self.process_frame(X[:, t, :], angle_rad, method)
            out.append(np.atleast_1d(self.transformer.istft(Yt[:, None, None])))
        return np.concatenate(out)


# --------------------------------------------------------------------------------------------
# adaptivefilter/SubbandAF.py, SubbandLMS.py, SubbandLmsMc.py, SubbandRLS.py (frequency-domain input)
# --------------------------------------------------------------------------------------------
class OracleSubbandLMS:
    """SubbandLMS.update with complex [K] inputs — SubbandAF.py:12-111, SubbandLMS.py:28-84."""

    def __init__(self, filter_len=2, num_bands=512, mu=0.1, normalization=True, alpha=0.9):
        self.N = filter_len
        self.half_band = int(num_bands / 2) + 1
        K = self.half_band
        self.W = np.zeros((K, filter_len), dtype=complex)
        self.mu = mu
        self.norm = normalization
        self.alpha = alpha
        self.input_buffer = np.zeros((K, filter_len), dtype=complex)
        self.P = np.zeros(K)

    def update(self, x_n, d_n, alpha=1e-4, p=None):
        K = self.half_band
        p = np.ones(K) if p is None else (np.full(K, p) if np.isscalar(p) else np.asarray(p).reshape(K))
        self.input_buffer[:, 1:] = self.input_buffer[:, :-1].copy()         # SubbandAF.py:50-51
        self.input_buffer[:, 0] = x_n
        out = np.einsum("ij,ij->i", self.W.conj(), self.input_buffer)       # SubbandAF.py:107
        err = d_n - out * p                                                  # SubbandLMS.py:66-68
        if self.norm:
            self.P = self.alpha * self.P + (1 - self.alpha) * np.sum(
                self.input_buffer.conj() * self.input_buffer, axis=-1).real  # :72-75
            grad = self.input_buffer * err[:, None].conj() / (self.P[:, None] + alpha)
        else:
            grad = self.input_buffer * err[:, None].conj()
        self.W = self.W + 2 * self.mu * grad * p[:, None]                    # SubbandAF.py:86
        return err, self.W


class OracleSubbandLmsMc:
    """SubbandLmsMc.update with complex [K, M] inputs — SubbandLmsMc.py:144-191."""

    def __init__(self, filter_len=2, num_bands=512, channel=4, mu=0.1, normalization=True, alpha=0.9):
        self.N, self.M = filter_len, channel
        self.half_band = int(num_bands / 2) + 1
        K = self.half_band
        self.W = np.zeros((K, filter_len, channel), dtype=complex)
        self.mu, self.norm, self.alpha = mu, normalization, alpha
        self.input_buffer = np.zeros((K, filter_len, channel), dtype=complex)
        self.P = np.zeros(K)

    def update(self, x_n, d_n, alpha=1e-4, p=None):
        K = self.half_band
        self.input_buffer[:, 1:, :] = self.input_buffer[:, :-1, :].copy()
        self.input_buffer[:, 0, :] = x_n
        out = np.einsum("ijk,ijk->i", self.W.conj(), self.input_buffer)
        pv = None if p is None else np.asarray(p).reshape(K)
        err = d_n - out * pv if pv is not None else d_n - out               # :168-172
        if self.norm:
            self.P = self.alpha * self.P + (1 - self.alpha) * np.einsum(
                "ijk,ijk->i", self.input_buffer.conj(), self.input_buffer).real / self.M    # :174-180
            grad = self.input_buffer * err[:, None, None].conj() / (self.P[:, None, None] + alpha)
        else:
            grad = self.input_buffer * err[:, None, None].conj()
        if pv is not None:
            self.W = self.W + 2 * self.mu * grad * pv[:, None, None]        # :136-137
        else:
            self.W = self.W + 2 * self.mu * grad
        return err, self.W


class OracleSubbandRLS:
    """SubbandRLS.update with complex [K] inputs — SubbandRLS.py:12-71."""

    def __init__(self, filter_len=2, num_bands=512, forgetting_factor=0.998, mu=0.5):
        self.N = filter_len
        self.half_band = int(num_bands / 2) + 1
        K = self.half_band
        self.W = np.zeros((K, filter_len), dtype=complex)
        self.mu = mu
        self.lam = forgetting_factor
        self.lam_inv = 1.0 / forgetting_factor
        self.input_buffer = np.zeros((K, filter_len), dtype=complex)
        self.P = np.tile(np.eye(filter_len, dtype=complex) / 1e-3, (K, 1, 1))   # :40-42

    def update(self, x_n, d_n, p=None):
        self.input_buffer[:, 1:] = self.input_buffer[:, :-1].copy()
        self.input_buffer[:, 0] = x_n
        X = self.input_buffer
        out = np.einsum("ij,ij->i", self.W.conj(), X)
        err = d_n - out                                                      # :52
        num = (self.P @ X[:, :, None])[..., 0]                               # :55
        kn = num / (self.lam + np.sum(X.conj() * num, axis=-1, keepdims=True))   # :56-60
        self.P = (self.P - kn[..., None] @ X[:, None, :].conj() @ self.P) * self.lam_inv   # :63
        self.W = self.W + 2 * self.mu * (err[:, None].conj() * kn)           # :65-66
        return err, self.W


# --------------------------------------------------------------------------------------------
# dereverberation/awpe.py — PARITY UNPINNED BY THE REFERENCE AS SHIPPED: Wpe.update calls an undefined
# check_input_data (awpe.py:150) and is built on the Subband filterbank.  This restates the equations on the
# STFT (Transform) grid with check_input_data(xd, x) := (analysis(xd), analysis(x)), return_td = True
# (the analogue of SubbandAF.update_input_data, SubbandAF.py:53-60) — SURVEY §8a-21's build decision.  The golden
# vectors (g10) come from the reference with exactly these two patches applied (make_golden.py R6, R7).
# --------------------------------------------------------------------------------------------
class OracleWpe:
    """Wpe.update — dereverberation/awpe.py:28-192 (on the STFT grid, see above)."""

    def __init__(self, channels=2, filter_len=2, num_bands=512, forgetting_factor=0.998, delay=4, hop_length=None):
        self.C, self.N = channels, filter_len
        self.half_band = int(num_bands / 2) + 1
        self.hop = int(num_bands / 2) if hop_length is None else hop_length
        K, C, N = self.half_band, channels, filter_len
        self.input_buffer = np.zeros((K, C, N), dtype=complex)                    # :58
        self.W = np.zeros((K, C, C * N), dtype=complex)                           # :61
        self.lam, self.lam_inv = forgetting_factor, 1.0 / forgetting_factor
        self.P = np.tile(np.eye(C * N, dtype=complex) * 1e-3, (K, 1, 1))          # :69-73
        self.D = delay
        self.delay_buf = np.zeros((delay * self.hop, C))                          # DelaySamples(hop, D*hop) :75-76
        self.var = np.zeros((K, 1))
        self.transform_x = OracleTransform(channel=C, n_fft=num_bands, hop_length=self.hop)
        self.transform_d = OracleTransform(channel=C, n_fft=num_bands, hop_length=self.hop)

    def update_fd(self, x_delayed, d_n):
        """frequency-domain core: x_delayed, d_n [K, C] -> err [K, C]  (:152-189)."""
        K, C, N = self.half_band, self.C, self.N
        if N > 1:
            self.input_buffer[:, :, 1:] = self.input_buffer[:, :, :-1].copy()     # :96-100
        self.input_buffer[:, :, 0] = x_delayed
        X = np.reshape(self.input_buffer, (K, -1))                                # :154
        err = d_n - np.einsum('kmi, ki->km', self.W.conj(), X)                    # :156-159
        var_n = np.abs(np.einsum('ij, ij->i', d_n.conj(), d_n)) / C               # :162
        self.var = 0.98 * self.var + (1 - 0.98) * var_n[:, None]                  # :163
        num = np.einsum('kij, kj->ki', self.P, X)                                 # :172
        kn = num / (self.lam * self.var + np.sum(X.conj() * num, axis=-1, keepdims=True))   # :173-178
        self.P = (self.P - np.einsum('ij,il,ilk->ijk', kn, X.conj(), self.P, optimize=True)) * self.lam_inv   # :181-183 (optimize=True as the reference: pairwise contraction)
        for ch in range(C):
            self.W[:, ch, :] = self.W[:, ch, :] + err[:, ch:ch + 1].conj() * kn   # :186-187
        return err

    def update(self, x_n):
        """x_n [hop, C] float -> dereverberated channel 0 [hop]  (:129-192)."""
        x_n = np.asarray(x_n, dtype=np.float64)
        buf = np.vstack((self.delay_buf, x_n))
        xd = buf[: x_n.shape[0]].copy()                                           # delayed by D*hop samples
        self.delay_buf = buf[x_n.shape[0]:].copy()
        Xd = self.transform_x.stft(xd)[:, 0, :]
        Dn = self.transform_d.stft(x_n)[:, 0, :]
        err = self.update_fd(Xd, Dn)
        return np.atleast_1d(self.transform_d.istft(err[:, 0])), self.W


# --------------------------------------------------------------------------------------------
# adaptivefilter/BaseFilter.py, RLS.py — sample-wise time-domain definitions
# --------------------------------------------------------------------------------------------
class OracleNlms:
    """BaseFilter.update — adaptivefilter/BaseFilter.py:25-85."""

    def __init__(self, filter_len=1024, mu=0.1, normalization=True):
        self.w = np.zeros(filter_len)
        self.buf = np.zeros(filter_len)
        self.mu, self.norm = mu, normalization

    def update(self, x_n, d_n, eps=1e-4, p=1.0):
        self.buf[1:] = self.buf[:-1].copy()                                  # :43-44
        self.buf[0] = x_n
        err = d_n - self.w @ self.buf                                         # :73
        grad = self.buf * err / (self.buf @ self.buf + eps) if self.norm else self.buf * err   # :75-78
        self.w = self.w + 2 * p * self.mu * grad                              # :82
        return err, self.w


class OracleRls:
    """Rls.update — adaptivefilter/RLS.py:14-42."""

    def __init__(self, filter_len=1024, mu=0.5, forgetting_factor=0.9998, delta=1e-3):
        self.w = np.zeros(filter_len)
        self.buf = np.zeros(filter_len)
        self.mu = mu
        self.P = np.eye(filter_len) / delta                                   # :20
        self.lam = forgetting_factor

    def update(self, x_n, d_n):
        self.buf[1:] = self.buf[:-1].copy()
        self.buf[0] = x_n
        err = d_n - self.w @ self.buf                                         # :30
        num = self.P @ self.buf                                               # :33
        kn = num / (self.lam + self.buf @ num)                                # :34
        self.P = (self.P - np.outer(kn, self.buf @ self.P)) / self.lam        # :37
        self.w = self.w + 2 * self.mu * err * kn                              # :39-40
        return err, self.w


# --------------------------------------------------------------------------------------------
# front-end conditioning + SubbandGSC (config 5 structure)
# --------------------------------------------------------------------------------------------
def fractional_delay_filter_bank(delays):
    """windowed-sinc fractional delay bank [filter_len, chs] — transform/multirate.py:4-51."""
    delays = np.array(delays, dtype=float)
    delays -= delays.min()
    N, Lw = delays.shape[0], 81
    filter_length = Lw + int(np.ceil(delays).max())
    bank = np.zeros((N, filter_length))
    di = np.floor(delays).astype(np.int64)
    df = delays - di
    T = np.arange(Lw)
    for i in range(N):
        bank[i, di[i]:di[i] + Lw] = np.hanning(Lw) * np.sinc(T - df[i] - (Lw - 1) / 2)
    return bank.T


class OracleDcNotch:
    """FilterDcNotch16.filter_dc_notch16 — adaptivefilter/feature.py:32-49."""

    def __init__(self, radius=0.9):
        self.radius = radius
        self.mem = np.zeros(2)

    def filter(self, x):
        r = self.radius
        den2 = r * r + 0.7 * (1 - r) * (1 - r)
        out = np.zeros(len(x))
        m0, m1 = self.mem
        for i in range(len(x)):
            vin = x[i]
            vout = m0 + vin
            m0 = m1 + 2 * (-vin + r * vout)
            m1 = vin - den2 * vout
            out[i] = r * vout
        self.mem = np.array([m0, m1])
        return out


class OracleTimeAlignment:
    """TimeAlignment (fractional-delay FIR pre-steering) — beamformer/fixedbeamformer.py:13-93."""

    def __init__(self, mic, angle_rad, fs=16000):
        tau = compute_tau(mic, angle_rad)                                       # :65 (MicArray.compute_tau)
        tau = -(tau - np.max(tau))                                              # :67
        self.delay_filter = fractional_delay_filter_bank(np.array(tau)[:, 0] * fs)   # :68-70
        self.L = self.delay_filter.shape[0]
        self.cache = np.zeros((self.L - 1, mic.M))

    def process(self, x):
        """x [samples, chs] -> [samples, chs]  (fir_filter :13-48: causal FIR with carried history)."""
        full = np.vstack((self.cache, x))
        out = np.zeros(x.shape)
        for m in range(x.shape[1]):
            out[:, m] = np.convolve(full[:, m], self.delay_filter[:, m])[self.L - 1: self.L - 1 + x.shape[0]]
        self.cache = full[-(self.L - 1):, :].copy()
        return out


class OracleDelaySamples:
    """DelaySamples — beamformer/utils.py:241-274."""

    def __init__(self, data_len, delay, channel=1):
        self.n_delay = delay
        self.buffer = np.zeros((data_len + delay, channel))

    def delay(self, x):
        if x.ndim == 1:
            x = x[:, None]
        n = x.shape[0]
        if self.n_delay == 0:
            return x
        self.buffer[-n:, :] = x
        out = self.buffer[:n, :].copy()
        self.buffer[: self.n_delay, :] = self.buffer[-self.n_delay:, :]
        return out


class OracleSubbandGSC:
    """SubbandGSC.process — beamformer/SubbandGSC.py:67-262.
    `rls_bm=True` is the config-5 composition of SURVEY section 8a-19: the adaptive blocking filters are
    SubbandRLS(filter_len=2, num_bands=2*frameLen) instead of SubbandLMS (a composition we define)."""

    def __init__(self, mic, frameLen=256, angle_deg=(197, 0), rls_bm=False):
        self.M, self.frameLen = mic.M, frameLen
        nb, hop, M = 2 * frameLen, frameLen, mic.M
        self.angle = np.array(angle_deg) / 180 * np.pi
        self.time_alignment = OracleTimeAlignment(mic, self.angle)               # :85
        self.rls_bm = rls_bm
        if rls_bm:
            self.bm = [OracleSubbandRLS(filter_len=2, num_bands=nb) for _ in range(M)]
        else:
            self.bm = [OracleSubbandLMS(filter_len=2, num_bands=nb, mu=1e-1) for _ in range(M)]      # :99-101
        self.bm_tx = [OracleTransform(n_fft=nb, hop_length=hop) for _ in range(M)]
        self.bm_td = [OracleTransform(n_fft=nb, hop_length=hop) for _ in range(M)]
        self.aic = OracleSubbandLmsMc(filter_len=2, num_bands=nb, channel=M, mu=0.01, alpha=0.8)     # :103-109
        self.aic_tx = OracleTransform(n_fft=nb, hop_length=hop, channel=M)
        self.aic_td = OracleTransform(n_fft=nb, hop_length=hop)
        self.delay_fbf = OracleDelaySamples(frameLen, frameLen)                  # :111
        self.spp = OracleMcSpp(nfft=nb, channels=M)                              # :115
        self.transform = OracleTransform(n_fft=nb, hop_length=hop, channel=M)    # :117
        self.dc_notch = [OracleDcNotch(radius=0.98) for _ in range(M)]           # :122-124
        self.omlsa_multi = OracleOmlsaMulti(nfft=nb, cal_weights=True, M=M)      # :127 (touched by postfilter=True only; output-dead)
        self.transform_fbf = OracleTransform(n_fft=nb, hop_length=hop)           # :128
        self.transform_bm = OracleTransform(n_fft=nb, hop_length=hop, channel=M) # :129

    def process(self, x, postfilter=False):
        """x [M, L] -> (output [L], fix_output [L], bm_output [L, M], p [K, blocks], aligned_output [L, M]).
        postfilter=True (:236-249) changes none of the five: it analyses the block's output and — as the reference does — the WHOLE
        bm_output array of the call as filled so far, takes frame 0 of that, and runs omlsa_multi.estimation on the two powers."""
        x = np.array(x, dtype=np.float64)
        M, FL = self.M, self.frameLen
        for m in range(M):
            x[m, :] = self.dc_notch[m].filter(x[m, :])                           # :177-178
        nblk = x.shape[1] // FL
        output = np.zeros(x.shape[1]); fix_output = np.zeros(x.shape[1])
        bm_output = np.zeros((x.shape[1], M)); aligned = np.zeros((x.shape[1], M))
        p = np.zeros((self.spp.half_bin, nblk))
        for n in range(nblk):
            sl = slice(n * FL, (n + 1) * FL)
            xa = self.time_alignment.process(x[:, sl].T)                         # :201
            aligned[sl] = xa
            D = self.transform.stft(xa)                                          # :204
            fixed = np.mean(xa, axis=1, keepdims=True)                           # :206
            with np.errstate(all="ignore"):
                p[:, n] = self.spp.estimation(D[:, 0, :])                        # :208
            for m in range(M):                                                   # :217-223
                X = self.bm_tx[m].stft(fixed[:, 0])[:, 0, 0]
                Dm = self.bm_td[m].stft(xa[:, m])[:, 0, 0]
                if self.rls_bm:
                    err, _ = self.bm[m].update(X, Dm)
                else:
                    err, _ = self.bm[m].update(X, Dm, p=p[:, n])
                bm_output[sl, m] = self.bm_td[m].istft(err)
            fixed_d = self.delay_fbf.delay(fixed)                                # :226
            Xa = self.aic_tx.stft(bm_output[sl, :])                              # :230-234 (x_n -> [K, 1, C])
            Dd = self.aic_td.stft(fixed_d[:, 0])[:, 0, 0]
            err, _ = self.aic.update(Xa[:, 0, :], Dd, p=1 - p[:, n])
            output[sl] = self.aic_td.istft(err)
            fix_output[sl] = fixed_d[:, 0]
            if postfilter:                                                       # :236-249
                Y = self.transform_fbf.stft(output[sl])                          # [K, 1, 1]
                U = self.transform_bm.stft(bm_output)                            # the whole array, every block: [K, nblk, M]
                self.omlsa_multi.estimation(np.real(Y[:, 0, 0] * np.conj(Y[:, 0, 0])), np.real(U[:, 0, :] * np.conj(U[:, 0, :])))
        return output, fix_output, bm_output, p, aligned


# --------------------------------------------------------------------------------------------
# overlap-save frequency-domain adaptive filters and the two GSCs built on them
# --------------------------------------------------------------------------------------------
class OracleFastFreqLms:
    """Overlap-save FDAF — adaptivefilter/FastFreqLms.py:48-245 (two_path: the foreground / background pair of :94-104,162-176).
    kind "plain" = FastFreqLms.update (:204-245); "bm" = AdaptiveBlockingMatrixFilter.update (gsc_bm.py:61-122,
    coefficient-clamped); "aic" = AdaptiveInterferenceCancellation.update (gsc_aic.py:53-108, norm-limited)."""

    def __init__(self, filter_len=128, mu=0.01, constrain=True, n_channels=1, alpha=0.9, non_causal=False,
                 kind="plain", weight_norm=False, two_path=False):
        self.filter_len, self.mu, self.constrain, self.n_channels, self.alpha = filter_len, mu, constrain, n_channels, alpha
        self.two_path = two_path
        self.hop_len, self.win_len = filter_len, 2 * filter_len                      # :62-63
        self.input_buffer = np.zeros((self.win_len, n_channels))                     # :65
        self.n_fft = 2 ** (int(np.log2(self.hop_len + filter_len - 1)) + 1)           # :70-71
        self.overlap = self.win_len - self.hop_len
        self.W = np.zeros((self.n_fft // 2 + 1, n_channels), dtype=complex)           # :76
        self.foreground = np.zeros_like(self.W)                                       # :95-96
        self.window = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(self.n_fft) / self.n_fft))[:, None]   # :91-92
        self.w = np.zeros((filter_len, n_channels))
        self.P = np.zeros((self.n_fft // 2 + 1, 1))                                   # :78
        self.non_causal = non_causal
        self.delay_samples = OracleDelaySamples(filter_len, filter_len // 2) if non_causal else None   # :85
        self.kind, self.weight_norm = kind, weight_norm
        n = self.n_fft
        self.m_upper = np.full(n // 2, 0.001)                                         # gsc_bm.py:48-59
        self.m_lower = np.full(n // 2, -0.001)
        self.m_upper[n // 4] = 0.9
        self.m_upper[[n // 4 + 1, n // 4 - 1]] = 0.3
        self.m_upper[[n // 4 + 2, n // 4 - 2]] = 0.05

    def update(self, x, d, p=1.0, fir_truncate=None, update=True):
        """x [hop] or [hop, C]; d [hop] or [hop, 1]; p scalar or [K, 1] -> (e [hop, 1], w [filter_len, C])."""
        x = np.asarray(x, dtype=np.float64)
        d = np.asarray(d, dtype=np.float64)
        if x.ndim == 1:
            x = x[:, None]
        n, hop = self.n_fft, self.hop_len
        self.input_buffer[: self.overlap] = self.input_buffer[-self.overlap:]         # :130
        self.input_buffer[-hop:] = x                                                  # :131
        X = np.fft.rfft(self.input_buffer, n=n, axis=0)                               # :155
        self.P = self.alpha * self.P + (1 - self.alpha) * np.sum(np.real(X.conj() * X), axis=1, keepdims=True)   # :156
        y = np.fft.irfft(np.sum(X * self.W, axis=-1))[-hop:][:, None]                 # :159-160
        if self.non_causal:
            d = self.delay_samples.delay(d)                                           # :167-168
        if d.ndim == 1:
            d = d[:, None]
        e = d - y                                                                     # :172
        if self.two_path:                                                             # :162-164,174-176
            y_f = np.sum(np.fft.irfft(X * self.foreground, axis=0)[-self.filter_len:, :], axis=1, keepdims=True)
            e_f = d - y_f
            if 10 * np.log10(np.sum(np.abs(e_f)) / (np.sum(np.abs(e)) + 1e-6) + 1e-6) > 3:      # transfer_logic :99-104
                self.foreground[:] = self.W
                y_f = self.window[self.filter_len:] * y_f + self.window[: self.filter_len] * y
            e = d - y_f
        E = np.fft.rfft(np.concatenate((np.zeros((self.overlap, 1)), e), axis=0), n=n, axis=0)   # :183-185
        self.P[self.P < 1e-4] = 1e-4                                                  # :187
        grad = X.conj() * E / self.P                                                  # :188
        norm = 1.0
        if self.kind == "plain":
            if self.constrain:                                                        # :194-198
                g = np.fft.irfft(grad, n=n, axis=0)
                g[-hop:] = 0
                grad = np.fft.rfft(g, n=n, axis=0)
            if update:
                self.W = self.W + p * 2 * self.mu * grad                              # :235
        else:
            if update:
                self.W = self.W + p * self.mu * grad                                  # gsc_bm.py:91 / gsc_aic.py:79
            if self.kind == "aic" and self.weight_norm:                               # gsc_aic.py:81-88
                nrm = np.sum(np.abs(self.W) ** 2) / n / n
                norm = np.sqrt(0.003 / nrm) if nrm > 0.003 else 1.0
            if self.constrain:
                w = np.fft.irfft(self.W, n=n, axis=0) * norm                          # gsc_aic.py:93 / gsc_bm.py:94
                w[-hop:] = 0
                if self.kind == "bm":                                                 # gsc_bm.py:97-108: every tap below n/2
                    w[: n // 2] = np.minimum(np.maximum(w[: n // 2], self.m_lower[:, None]), self.m_upper[:, None])
                self.W = np.fft.rfft(w, n=n, axis=0)
        self.w = np.fft.irfft(self.W, n=n, axis=0)[: self.filter_len, :]              # :237-238
        if fir_truncate is not None:                                                  # :240-244
            ws = self.w.copy()
            ws[:fir_truncate] = 0.0
            ws[-fir_truncate:] = 0.0
            self.W = np.fft.rfft(ws * norm, n=n, axis=0)
        return e, self.w


def _td_blocking_matrix(x):
    """adjacent-pair differences — TDGSC.py:69-87."""
    return x[:, :-1] - x[:, 1:]


class OracleTDGSC:
    """TDGSC.process — beamformer/TDGSC.py:24-175."""

    def __init__(self, mic, frameLen=256, angle_deg=(197, 0)):
        self.M, self.frameLen = mic.M, frameLen
        nb, M = 2 * frameLen, mic.M
        self.time_alignment = OracleTimeAlignment(mic, np.array(angle_deg) / 180 * np.pi)      # :36
        self.aic_filter = OracleFastFreqLms(filter_len=frameLen, n_channels=M - 1, non_causal=True)   # :37
        self.dc_notch = [OracleDcNotch(radius=0.98) for _ in range(M)]                # :38-40
        self.spp = OracleMCRA(nfft=nb, L=65)                                          # :42-43,46
        self.transform = OracleTransform(n_fft=nb, hop_length=frameLen)               # :44
        self.omlsa_multi = OracleOmlsaMulti(nfft=nb, cal_weights=True, M=M)           # :48
        self.transform_fbf = OracleTransform(n_fft=nb, hop_length=frameLen)           # :49
        self.transform_bm = OracleTransform(n_fft=nb, hop_length=frameLen, channel=M - 1)   # :50

    def process(self, x, postfilter=False):
        """x [samples, chs] -> (output [samples], p [K, blocks], output_bm [samples, chs-1])."""
        x = np.array(x, dtype=np.float64)
        M, FL = self.M, self.frameLen
        for m in range(M):
            x[:, m] = self.dc_notch[m].filter(x[:, m])                                # :129-130
        nblk = x.shape[0] // FL
        output = np.zeros(x.shape[0]); output_bm = np.zeros((x.shape[0], M - 1))
        p = np.zeros((self.spp.half_bin, nblk))
        for n in range(nblk):
            sl = slice(n * FL, (n + 1) * FL)
            xa = self.time_alignment.process(x[sl])                                   # :143 -> :66
            fixed = np.mean(xa, axis=1, keepdims=True)                                # :67
            D = self.transform.stft(fixed)                                            # :145
            self.spp.estimation(D[:, 0, :])                                           # :146
            p[:, n] = self.spp.p
            bm = _td_blocking_matrix(xa)                                              # :149
            out_n, _ = self.aic_filter.update(bm, fixed, fir_truncate=30, p=1 - p[:, n:n + 1])   # :152-156 -> :105
            if postfilter:                                                            # :158-170
                Y = self.transform_fbf.stft(out_n)
                U = self.transform_bm.stft(bm)
                self.omlsa_multi.estimation(np.real(Y[:, 0, 0] * np.conj(Y[:, 0, 0])), np.real(U[:, 0, :] * np.conj(U[:, 0, :])))
                Y[:, 0, 0] = Y[:, 0, 0] * np.sqrt(self.omlsa_multi.G)
                out_n = self.transform_fbf.istft(Y)
            output_bm[sl] = bm
            output[sl] = np.squeeze(out_n)
        return output, p, output_bm


class OracleFDGSC:
    """FDGSC.process (blocking-matrix mode 3) — beamformer/FDGSC.py:38-317."""

    def __init__(self, mic, frameLen=256, angle_deg=(197, 0)):
        self.M, self.frameLen = mic.M, frameLen
        nb, M = 2 * frameLen, mic.M
        self.time_alignment = OracleTimeAlignment(mic, np.array(angle_deg) / 180 * np.pi)      # :56
        self.bm = [OracleFastFreqLms(filter_len=frameLen, mu=0.1, alpha=0.9, kind="bm") for _ in range(M)]   # :71-81
        self.aic_filter = OracleFastFreqLms(filter_len=frameLen, n_channels=M, mu=0.1, alpha=0.9, kind="aic",
                                            weight_norm=True)                         # :83-91
        self.delay_fbf = OracleDelaySamples(frameLen, frameLen)                       # :93
        self.delay_aligned = OracleDelaySamples(frameLen, frameLen // 2, channel=M)   # :96
        self.spp = OracleMCRA(nfft=nb, L=60)                                          # :99-100
        self.transform_x = OracleTransform(n_fft=nb, hop_length=frameLen, channel=M)  # :106
        self.omlsa_multi = OracleOmlsaMulti(nfft=nb, cal_weights=True, M=M)           # :108
        self.transform_fbf = OracleTransform(n_fft=nb, hop_length=frameLen)           # :109
        self.transform_bm = OracleTransform(n_fft=nb, hop_length=frameLen, channel=M - 1)   # :110
        self.dc_notch = [OracleDcNotch(radius=0.98) for _ in range(M)]                # :114-116

    def process(self, x, postfilter=False, dc_notch=True):
        """x [samples, chs] -> (output, p, fix_output, fix_output_delayed, bm_output, aligned, aligned_delayed)."""
        x = np.array(x, dtype=np.float64)
        M, FL = self.M, self.frameLen
        if dc_notch:
            for m in range(M):
                x[:, m] = self.dc_notch[m].filter(x[:, m])                            # :213-215
        ns = x.shape[0]
        nblk = ns // FL
        output = np.zeros(ns); bm_output = np.zeros((ns, M))
        aligned = np.zeros((ns, M)); aligned_d = np.zeros((ns, M))
        fix = np.zeros(ns); fix_d = np.zeros(ns)
        p = np.zeros((self.spp.half_bin, nblk))
        for n in range(nblk):
            sl = slice(n * FL, (n + 1) * FL)
            xa = self.time_alignment.process(x[sl])                                   # :235
            fixed = np.mean(xa, axis=1, keepdims=True)                                # :238
            D = self.transform_x.stft(x[sl])                                          # :241
            self.spp.estimation(D[:, 0, :])                                           # :243 (channel 0 only, mcra.py:32-33)
            p[:, n] = self.spp.p
            if np.mean(p[32:128, n]) > 0.8:                                           # :248-255
                lo = p[:32, n]
                lo[lo < 0.8] = 0.8
            xad = self.delay_aligned.delay(xa)                                        # :258
            bm_n = np.zeros((FL, M))
            for m in range(M):                                                        # :259-264 -> :185-195 (mode 3, p = 1.0)
                e, _ = self.bm[m].update(fixed, xad[:, m], p=1.0)
                bm_n[:, m] = e[:, 0]
            bm_output[sl] = bm_n
            fixed_dn = self.delay_fbf.delay(fixed)                                    # :270
            Y = self.transform_fbf.stft(fixed_dn)                                     # :273 (advances the shared transform_fbf state)
            out_n, _ = self.aic_filter.update(bm_n, fixed_dn, p=1 - np.mean(p[:, n]))   # :278-284
            if postfilter:                                                            # :286-298
                Y = self.transform_fbf.stft(out_n)
                U = self.transform_bm.stft(bm_output[:, :-1])                         # :288 — the WHOLE array, every block
                self.omlsa_multi.estimation(np.real(Y[:, 0, 0] * np.conj(Y[:, 0, 0])), np.real(U[:, 0, :] * np.conj(U[:, 0, :])))
                Y[:, 0, 0] = Y[:, 0, 0] * np.sqrt(self.omlsa_multi.G)
                out_n = self.transform_fbf.istft(Y)
            fix[sl] = fixed[:, 0]; fix_d[sl] = fixed_dn[:, 0]
            aligned[sl] = xa; aligned_d[sl] = xad
            output[sl] = np.squeeze(out_n)
        return output, p, fix, fix_d, bm_output, aligned, aligned_d


# --------------------------------------------------------------------------------------------
# BASELINE config 4: WPE dereverberation -> adaptive MVDR -> SPP gain (a composition the reference never writes down)
# --------------------------------------------------------------------------------------------
class OracleWpeMvdrPostfilter:
    """Config-4 composition, defined here from the reference's own pieces because no reference class composes them
    (parity of the composition is therefore UNPINNED by the reference; each piece is pinned: Wpe G10 (patched reference),
    adaptivebeamfomer G4, McMcra G5):
      D   = Transform.stft(x)                              transform.py:430-453
      E   = Wpe frequency-domain core on (D delayed by `delay` frames, D), all C channels   awpe.py:152-189
      G   = McMcra.estimation(E).G                          mc_mcra.py:179-224 (the GSC post-filter convention, GSC.py:225,286)
      Y   = adaptivebeamfomer frame loop on E (MCRA-gated Rvv, MVDR weights) * G          adaptivebeamformer.py:69-120
      out = Transform.istft(Y)                              transform.py:455-481"""

    def __init__(self, mic, nfft=1024, hop=512, taps=2, delay=4, mcra_L=15):
        M = mic.M
        self.M, self.nfft, self.hop = M, nfft, hop
        self.tf = OracleTransform(channel=M, n_fft=nfft, hop_length=hop)
        self.wpe = OracleWpe(channels=M, filter_len=taps, num_bands=nfft, delay=delay, hop_length=hop)
        self.mvdr = OracleAdaptiveMVDR(mic, frameLen=nfft, hop=hop, nfft=nfft, mcra_L=mcra_L)
        self.spp = OracleMcMcra(nfft=nfft, channels=M)
        self.ring = [np.zeros((nfft // 2 + 1, M), dtype=complex) for _ in range(delay)]

    def process(self, x, angle_rad, method=2):
        """x [M, T*hop] -> y [T*hop]."""
        D = self.tf.stft(np.asarray(x).T)
        out = []
        for t in range(D.shape[1]):
            self.ring.append(D[:, t, :])
            Xd = self.ring.pop(0)
            E = self.wpe.update_fd(Xd, D[:, t, :])
            self.spp.estimation(E)
            Y = self.mvdr.process_frame(E, angle_rad, method) * self.spp.G
            out.append(np.atleast_1d(self.tf.istft(Y[:, None, None])))
        return np.concatenate(out)


class OracleMvdrPostfilter:
    """MVDR + post-filter in one pass (BASELINE.json north_star's target workload; DS_ALGO_ADAPTIVE_PF).  No reference class composes the
    two; the composition is GSC.process's own convention for its beamformer and `spp` (GSC.py:225,286), on the adaptive beamformer:
      Z   = Transform.stft(x)                               transform.py:430-453
      G   = McMcra.estimation(Z).G                          mc_mcra.py:179-224  (GSC.py:225)
      Y   = adaptivebeamfomer frame loop on Z (MCRA-gated Rvv, src / DS / MVDR weights) * G   adaptivebeamformer.py:69-120, GSC.py:286
      out = Transform.istft(Y)                              transform.py:455-481
    Pinned by the G23 fixtures: the same composition driven through the reference's own objects (tests/golden/make_golden.py g23)."""

    def __init__(self, mic, nfft=512, hop=None, mcra_L=15):
        self.mvdr = OracleAdaptiveMVDR(mic, frameLen=nfft, hop=hop, nfft=nfft, mcra_L=mcra_L)
        self.spp = OracleMcMcra(nfft=nfft, channels=mic.M)
        self.tf = self.mvdr.transformer

    def process(self, x, angle_rad, method=2):
        """x [M, T*hop] -> y [T*hop]; T successive one-hop calls."""
        D = self.tf.stft(np.asarray(x).T)
        out = []
        for t in range(D.shape[1]):
            self.spp.estimation(D[:, t, :])
            Y = self.mvdr.process_frame(D[:, t, :], angle_rad, method) * self.spp.G
            out.append(np.atleast_1d(self.tf.istft(Y[:, None, None])))
        return np.concatenate(out)


# --------------------------------------------------------------------------------------------
# synthetic input (SURVEY §8d / BASELINE.md §3) — shared by tests and the cpu_baseline leg
# --------------------------------------------------------------------------------------------
def synth_utterance(utt_index, n_samples, mic, angle_deg=(197, 0), fs=16000):
    """Seeded synthetic M-channel utterance [M, n_samples] float32: white noise sigma=0.05 per mic
    + 0.5 s on / 0.5 s off band-limited (300-3400 Hz) Gaussian source sigma=0.1 delayed per mic by
    the far-field delays of `angle_deg` (fractional delays applied as a phase ramp in the DFT domain)."""
    rng = np.random.default_rng(1234 + int(utt_index))
    M = mic.M
    noise = rng.standard_normal((M, n_samples)) * 0.05
    src = rng.standard_normal(n_samples)
    S = np.fft.rfft(src)
    f = np.fft.rfftfreq(n_samples, 1.0 / fs)
    S[(f < 300) | (f > 3400)] = 0
    gate = ((np.arange(n_samples) // (fs // 2)) % 2 == 0).astype(np.float64)
    tau = compute_tau(mic, np.array(angle_deg) / 180 * np.pi)[:, 0]
    x = np.empty((M, n_samples))
    for m in range(M):
        sm = np.fft.irfft(S * np.exp(-2j * np.pi * f * tau[m]), n_samples)
        sm = sm / (np.std(sm) + 1e-12) * 0.1
        x[m] = sm * gate + noise[m]
    return x.astype(np.float32)
