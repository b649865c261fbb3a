"""Frame-level objects with the reference's names and call contracts, backed by libdsenh.so kernels.

  Transform             stft / istft / analysis / synthesis   (transform/transform.py:407-496)
  NoiseEstimationMCRA   estimation(Y) -> lambda_d             (noise_estimation/mcra.py:20-77)
  McMcra                estimation(y) ; attrs p, G, xi, gamma (noise_estimation/mc_mcra.py:25-224)
  NsOmlsaMulti          estimation(y, u) -> lambda_d          (noise_estimation/omlsa_multi.py:27-156)
  SubbandLMS / SubbandLmsMc / SubbandRLS   update(x_n, d_n, p=) -> (err, W)
                                                              (adaptivefilter/SubbandLMS.py, SubbandLmsMc.py, SubbandRLS.py)
  Wpe                   update(x_n) -> (out, W)               (dereverberation/awpe.py:28-192, on the STFT grid)

Every class accepts ``batch=B`` (default 1 = the reference's shapes; B > 1 adds a leading batch axis).
All arithmetic runs on the GPU (fp32); the classes only reshape arrays between the reference's
layouts and the native [B][T][K][..] layout."""
import numpy as np

from . import _lib as L
from .engine import BatchEngine


class _Base(object):
    batch = 1

    def _sq(self, a):
        return a[0] if self.batch == 1 else a

    def _add_batch(self, a, ndim_single):
        a = np.asarray(a)
        if a.ndim == ndim_single:
            if self.batch != 1:
                raise ValueError("object built with batch=%d; pass arrays with a leading batch axis" % self.batch)
            a = a[None]
        return a


class Transform(_Base):
    """Streaming multichannel STFT / ISTFT with carried overlap — transform/transform.py:407-496."""

    def __init__(self, channel=1, n_fft=256, hop_length=128, window=None, batch=1, device=-1):
        if window is not None and np.asarray(window).shape != (n_fft,):
            raise NotImplementedError("a custom window must have n_fft samples (win_len == n_fft is what the kernels frame)")
        self.channel, self.n_fft, self.frame_length, self.hop_length = channel, n_fft, n_fft, hop_length
        self.half_bin = int(n_fft / 2 + 1)
        self.window = np.sqrt(0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)) if window is None else np.asarray(window, dtype=np.float64)
        self.win_len = n_fft
        self.overlap = n_fft - hop_length
        self.W0 = np.sum(self.window ** 2)
        self.batch = int(batch)
        self._eng = BatchEngine(L.ALGO_TRANSFORM, channel, n_fft, hop_length, batch=batch, device=device)
        if window is not None:
            self._eng.set_window(self.window)                              # transform.py:415-416

    def stft(self, x):
        """x [samples, channels] or [samples] -> [half_bin, frames, channels] complex128."""
        x = np.asarray(x)
        if x.ndim == 1 and self.batch == 1:
            x = x[:, None]
        x = self._add_batch(x, 2)
        if x.shape[1] % self.hop_length != 0:
            raise ValueError("samples (%d) must be a multiple of hop (%d)" % (x.shape[1], self.hop_length))
        Y = self._eng.stft(x, L.LAYOUT_SAMPLES_CHANNELS)                    # [B, T, K, C]
        return self._sq(np.transpose(Y, (0, 2, 1, 3)).astype(np.complex128))

    def istft(self, Y):
        """Y [half_bin], [half_bin, channels] (one frame) or [half_bin, frames, channels] -> [samples(, ch)]."""
        Y = np.asarray(Y)
        if self.batch == 1:
            if Y.ndim == 1:
                Y = Y[:, None, None]                                        # transform.py:461-462
            if Y.ndim == 2:
                Y = Y[:, None, :]                                           # :463-464: 2-D means [K, channels]
            Y = Y[None]
        half_bin, n_channels = Y.shape[1], Y.shape[3]
        assert n_channels <= self.channel, 'n_channels:{} != self.channel:{}'.format(n_channels, self.channel)
        y = self._eng.istft(np.transpose(Y, (0, 2, 1, 3)))                  # [B, L, C]
        out = y.astype(np.float64)
        return np.squeeze(out[0]) if self.batch == 1 else out

    def magphase(self, D, power=1):
        mag = np.abs(D) ** power
        return mag, np.exp(1.0j * np.angle(D))

    analysis = stft
    synthesis = istft

    @property
    def previous_input(self):
        """[overlap, channel] (transform.py:425,451): the last n_fft - hop input samples"""
        return self._sq(np.swapaxes(self._eng.get_field(L.FIELD_STFT_TAIL), 1, 2).astype(np.float64))

    @property
    def previous_output(self):
        """[overlap, channel] (transform.py:426,477): the synthesis frames' overlap still to be added to the next hops (before the
        hop / W0 scaling of :479, like the reference's attribute)"""
        return self._sq(np.swapaxes(self._eng.get_field(L.FIELD_OLA_TAIL), 1, 2).astype(np.float64))


class NoiseEstimationMCRA(_Base):
    """Minima-controlled recursive averaging — noise_estimation/mcra.py:20-77, NoiseEstimationBase.py:5-60."""

    def __init__(self, nfft=256, p_max=0.999, p_min=1e-3, batch=1, device=-1):
        self.nfft, self.half_bin, self.batch = nfft, int(nfft / 2 + 1), int(batch)
        self.p_max, self.p_min = p_max, p_min
        self._L = 15
        self._eng = BatchEngine(L.ALGO_MCRA, 1, nfft, batch=batch, device=device)

    @property
    def L(self):
        return self._L

    @L.setter
    def L(self, v):
        self._L = int(v)
        self._eng.set_mcra_L(int(v))

    def estimation(self, Y):
        """Y [half_bin] power (or complex, or [half_bin, ch] -> column 0) -> lambda_d [half_bin]."""
        Y = np.asarray(Y)
        if self.batch == 1:
            if Y.ndim > 1:
                Y = Y[:, 0]                                                 # mcra.py:32-33
            Y = Y[None]
        assert Y.shape[-1] == self.half_bin, 'len(Y):{} != half_bin:{}'.format(Y.shape[-1], self.half_bin)
        lam = self._eng.mcra_estimate(Y[:, None, :])                        # one frame
        return self._sq(lam[:, 0, :].astype(np.float64))

    def _row(self, f):
        return self._sq(self._eng.op_state()[:, f, :].astype(np.float64))

    S = property(lambda s: s._row(0))
    Smin = property(lambda s: s._row(1))
    Stmp = property(lambda s: s._row(2))
    p = property(lambda s: s._row(3))
    lambda_d = property(lambda s: s._row(4))
    frm_cnt = property(lambda s: int(s._eng.get_field(L.FIELD_COUNTERS)[0, 0]))


class McMcra(_Base):
    """Multichannel speech presence probability with real covariances — noise_estimation/mc_mcra.py:25-224."""

    def __init__(self, nfft=256, channels=4, batch=1, device=-1):
        self.nfft, self.half_bin, self.channels, self.M, self.batch = nfft, int(nfft / 2 + 1), channels, channels, int(batch)
        self._eng = BatchEngine(L.ALGO_MCMCRA, channels, nfft, batch=batch, device=device)
        self._ns = channels * (channels + 1) // 2

    def estimation(self, y):
        """y complex [half_bin, channels] (one frame); results in attributes p, G, xi, gamma."""
        y = self._add_batch(y, 2)
        self._eng.mcmcra_estimate(y[:, None, :, :])

    def _row(self, f):
        return self._sq(self._eng.op_state()[:, f, :].astype(np.float64))

    def _sym(self, base):
        st = self._eng.op_state()
        M = self.M
        out = np.zeros((self.batch, M, M, self.half_bin))
        q = 0
        for i in range(M):
            for j in range(i, M):
                out[:, i, j, :] = out[:, j, i, :] = st[:, base + q, :]
                q += 1
        return self._sq(out)                                               # reference layout [M, M, half_bin]

    Phi_yy = property(lambda s: s._sym(0))
    Phi_vv = property(lambda s: s._sym(s._ns))
    xi = property(lambda s: s._row(2 * s._ns + 0))
    gamma = property(lambda s: s._row(2 * s._ns + 1))
    p = property(lambda s: s._row(2 * s._ns + 2))
    G = property(lambda s: s._row(2 * s._ns + 3))
    frm_cnt = property(lambda s: int(s._eng.get_field(L.FIELD_COUNTERS)[0, 0]))


class McSppBase(_Base):
    """Multichannel SPP with MCRA prior + PMWF weights — noise_estimation/mcspp_base.py:28-324."""

    def __init__(self, nfft=256, channels=4, batch=1, device=-1):
        self.nfft, self.half_bin, self.channels, self.batch = nfft, int(nfft / 2 + 1), channels, int(batch)
        self._eng = BatchEngine(L.ALGO_MCSPPBASE, channels, nfft, batch=batch, device=device)
        self._o = 2 * channels * channels + 5

    def estimation(self, y):
        """y complex [half_bin, channels] -> p [half_bin]; PMWF weights in `w` [half_bin, channels]."""
        y = self._add_batch(y, 2)
        p, w = self._eng.mcsppbase_estimate(y[:, None, :, :])
        self._w = w[:, 0].astype(np.complex128)
        return self._sq(p[:, 0, :].astype(np.float64))

    def _row(self, f):
        return self._sq(self._eng.op_state()[:, f, :].astype(np.float64))

    def _herm(self, base):
        st = self._eng.op_state().astype(np.float64)
        M = self.channels
        out = np.zeros((self.batch, self.half_bin, M, M), dtype=complex)
        q = 0
        for i in range(M):
            out[:, :, i, i] = st[:, base + i, :]
        for i in range(M):
            for j in range(i + 1, M):
                v = st[:, base + M + 2 * q, :] + 1j * st[:, base + M + 2 * q + 1, :]
                out[:, :, i, j] = v
                out[:, :, j, i] = np.conj(v)
                q += 1
        return self._sq(out)                                               # [half_bin, M, M] like the reference

    Phi_yy = property(lambda s: s._herm(0))
    Phi_vv = property(lambda s: s._herm(s.channels * s.channels))
    xi = property(lambda s: s._row(s._o + 0))
    gamma = property(lambda s: s._row(s._o + 1))
    p = property(lambda s: s._row(s._o + 2))
    w = property(lambda s: s._sq(s._w))

    @property
    def Phi_vv_inv(self):
        """inv(Re(Phi_vv) + 1e-6 I) (mcspp_base.py:277-279), derived on the host on demand."""
        R = np.real(self._herm(self.channels * self.channels))
        return np.linalg.inv(R + np.eye(self.channels) * 1e-6).astype(complex)


class McSpp(_Base):
    """Multichannel SPP with a coherent-to-diffuse-ratio prior (McCDR) — noise_estimation/mcspp.py:46-305,
    mccdr.py:25-177.  estimation(y) returns p and leaves Phi_xx / Phi_vv_inv / w exactly as the reference does after the
    call; `mvdr_out` additionally holds the notebook's online-MVDR output for that frame (example/mvdr.ipynb cell 4:
    steering(Phi_xx) -> compute_mvdr_weight(steer, Phi_vv_inv) -> sum conj(w) y), computed in the same kernel.

    Like the reference, the McCDR prior hard-wires a circular r = 0.032 array of `channels` microphones and uses the
    pair (1, 2) (mccdr.py:63,141); unlike the reference (IndexError, mcspp.py:54) any supported channel count works."""

    @staticmethod
    def diffuse_coherence(channels, nfft):
        """Fn[K]: diffuse-field coherence of microphones 1, 2 of the circular r = 0.032 array McCDR hard-wires (mccdr.py:63,141)."""
        from .mic_array import MicArray, gen_noise_msc
        channels = getattr(channels, "M", channels)
        return gen_noise_msc(MicArray(arrayType="circular", r=0.032, M=channels), nfft)[:, 1, 2]

    def __init__(self, nfft=256, channels=4, mic_array=None, batch=1, device=-1):
        self.nfft, self.half_bin, self.channels, self.batch = nfft, int(nfft / 2 + 1), channels, int(batch)
        self._eng = BatchEngine(L.ALGO_MCSPP, channels, nfft, batch=batch, device=device)
        self._eng.set_aux(self.diffuse_coherence(channels, nfft))
        self.mic_array = mic_array
        if mic_array is not None:
            self.steer_vector = mic_array.steering_vector(look_direction=30).T          # mcspp.py:64-66
        self._last = None
        self._repeat = False
        self._o = 12 + 2 * channels * channels              # ds_ops.hpp MCSPP_ROW0: rows 0..8 McCDR, 9..11 unused, then the two matrices
        self.frm_cnt = 0

    def estimation(self, y, diag_value=1e-4, repeat=False):
        """y complex [half_bin, channels] -> p [half_bin]."""
        if bool(repeat) != self._repeat:                                     # mcspp.py:280-282: a second estimation_core after the noise update
            self._repeat = bool(repeat)
            self._eng.set_mcspp_repeat(self._repeat)
        y = self._add_batch(y, 2)
        self._last = self._eng.mcspp_estimate(y[:, None, :, :], want_yout=True, want_matrices=True)
        self.frm_cnt += 1
        return self.p

    def _get(self, key):
        if self._last is None:
            raise AttributeError("call estimation() first")
        return self._sq(self._last[key][:, 0])

    p = property(lambda s: s._get("p").astype(np.float64))
    w = property(lambda s: s._get("w_pmwf").astype(np.complex128))
    Phi_xx = property(lambda s: s._get("phi_xx").astype(np.complex128))
    Phi_vv_inv = property(lambda s: s._get("phi_vv_inv").astype(np.complex128))
    mvdr_out = property(lambda s: s._get("yout").astype(np.complex128))
    xi = property(lambda s: s._sq(s._eng.op_state()[:, s._o, :].astype(np.float64)))
    gamma = property(lambda s: s._sq(s._eng.op_state()[:, s._o + 1, :].astype(np.float64)))


class OnlineMvdr(_Base):
    """The online MVDR of example/mvdr.ipynb cell 4 in ONE native call (DS_ALGO_MCSPP_MVDR): `transform.stft` -> per frame
    `noise_estimator.estimation(y)` (McSpp with the McCDR prior, mcspp.py:244-305) -> `steering(noise_estimator.Phi_xx)`
    (beamformer.py:10-31) -> `compute_mvdr_weight(steer_vector, noise_estimator.Phi_vv_inv)` (beamformer.py:133-155) ->
    `Yout[:, n] = w^H y` -> `transform.istft(Yout)`.  No reference class wraps the cell; the constructor takes what the cell sets up
    (n_fft = 512, hop 256, the microphone count) and process() takes the array the cell hands to `transform.stft`.
    A call of T hops is T successive one-hop calls, state carried (analysis / synthesis overlaps, McSpp's matrices and counters)."""

    def __init__(self, nfft=512, hop_length=None, channels=6, repeat=False, batch=1, device=-1):
        self.nfft, self.hop, self.channels, self.batch = int(nfft), int(nfft // 2 if hop_length is None else hop_length), int(channels), int(batch)
        self.half_bin = self.nfft // 2 + 1
        self._eng = BatchEngine(L.ALGO_MCSPP_MVDR, channels, self.nfft, hop=self.hop, batch=batch, device=device)
        self._eng.chain_set_aux(L.CHAIN_AUX_COHERENCE, McSpp.diffuse_coherence(channels, self.nfft))
        if repeat:
            self._eng.set_mcspp_repeat(True)                                  # noise_estimator.estimation(y, repeat=True), mcspp.py:280-282
        self.p = None

    def process(self, x):
        """x [samples, channels] (or [B, samples, channels]) -> yout [samples]; self.p [half_bin, frames] as the cell collects it."""
        x = self._add_batch(x, 2)
        if x.shape[1] % self.hop != 0 or x.shape[2] != self.channels:
            raise ValueError("x must be [k * hop (%d) samples, %d channels]" % (self.hop, self.channels))
        y, p = self._eng.mcspp_mvdr_process(x, L.LAYOUT_SAMPLES_CHANNELS)
        self.p = self._sq(np.swapaxes(p, 1, 2).astype(np.float64))
        return self._sq(y.astype(np.float64))

    def reset(self):
        self._eng.reset()


_linalg_engines = {}


def _linalg(K, M, B):
    key = (K, M, B)
    if key not in _linalg_engines:
        _linalg_engines[key] = BatchEngine(L.ALGO_LINALG, M, 2 * (K - 1), batch=B)
    return _linalg_engines[key]


def steering(XXs):
    """principal eigenvector of each Hermitian matrix [bins, M, M] -> [bins, M], phase-normalised by the reference
    sensor — beamformer/beamformer.py:10-31 (batched complex Jacobi on the GPU)."""
    XXs = np.asarray(XXs)
    single = XXs.ndim == 3
    X = XXs[None] if single else XXs
    v = _linalg(X.shape[1], X.shape[2], X.shape[0]).steering(X).astype(np.complex128)
    return v[0] if single else v


def compute_mvdr_weight(steer_vector, Rvv_inv, Gmin=0.0631, beta=1):
    """w = R^-1 a / (a^H R^-1 a) per bin — beamformer/beamformer.py:133-155 (GPU)."""
    a, R = np.asarray(steer_vector), np.asarray(Rvv_inv)
    single = a.ndim == 2
    if single:
        a, R = a[None], R[None]
    w = _linalg(a.shape[1], a.shape[2], a.shape[0]).mvdr_weight(a, R).astype(np.complex128)
    return w[0] if single else w


def compute_pmwf_weight(xi, Rxx, Rvv_inv, Gmin=0.0631, beta=1):
    """parameterised multichannel Wiener filter w = (Rvv_inv Rxx) e_0 / (beta + xi) — beamformer/beamformer.py:100-130, with the
    matrices as [bins, M, M] (the reference's docstring says [M, M, bins] but its batched `Rvv_inv @ Rxx @ u` needs [bins, M, M], and
    its `channels = Rxx.shape[0]` only works where M == bins: here channels = Rxx.shape[1]).  xi [bins] -> w [bins, M] (GPU)."""
    xi, Rxx, Rvv_inv = np.asarray(xi), np.asarray(Rxx), np.asarray(Rvv_inv)
    single = Rxx.ndim == 3
    if single:
        xi, Rxx, Rvv_inv = xi[None], Rxx[None], Rvv_inv[None]
    w = _linalg(Rxx.shape[1], Rxx.shape[2], Rxx.shape[0]).pmwf_weight(xi, Rxx, Rvv_inv, beta).astype(np.complex128)
    return w[0] if single else w


def get_gev_vector(target_psd_matrix, noise_psd_matrix):
    """GEV beamforming vector: the principal generalised eigenvector of (target, noise) per bin, v^H N v = 1 —
    beamformer/beamformer.py:79-97 (scipy.linalg.eigh(a, b)[1][:, -1]).  An eigenvector's phase is the eigen-solver's business
    (LAPACK's in the reference; phase_correction removes it from bin to bin): here the first component of the whitened vector is real
    and positive.  [bins, M, M] -> [bins, M] (Cholesky whitening + complex Jacobi in double on the GPU)."""
    A, N = np.asarray(target_psd_matrix), np.asarray(noise_psd_matrix)
    single = A.ndim == 3
    if single:
        A, N = A[None], N[None]
    v = _linalg(A.shape[1], A.shape[2], A.shape[0]).gev_vector(A, N).astype(np.complex128)
    return v[0] if single else v


def blind_analytic_normalization(vector, noise_psd_matrix, eps=0):
    """vector * |sqrt(v^H N N v)| / (|v^H N v| + eps) — beamformer/beamformer.py:34-63.  [bins, M], [bins, M, M] -> [bins, M] (GPU)."""
    v, N = np.asarray(vector), np.asarray(noise_psd_matrix)
    single = v.ndim == 2
    if single:
        v, N = v[None], N[None]
    out = _linalg(v.shape[1], v.shape[2], v.shape[0]).blind_analytic_normalization(v, N, eps).astype(np.complex128)
    return out[0] if single else out


def phase_correction(vector):
    """bin f rotated by exp(-j angle(sum_m w[f, m] conj(w[f - 1, m]))), bins in order — beamformer/beamformer.py:66-76.  [bins, M] (GPU)."""
    v = np.asarray(vector)
    single = v.ndim == 2
    if single:
        v = v[None]
    out = _linalg(v.shape[1], v.shape[2], v.shape[0]).phase_correction(v).astype(np.complex128)
    return out[0] if single else out


def wpe_block_layout(C, N):
    """The per-bin state block of the RLS-WPE kernels (csrc/ds_wpe.hpp wpe_layout()): complex-word offsets of W (w0) and of the taps (x0),
    float offset of var, floats per block.  C N == 16 is laid out on 128-byte lines (the triangle's 136 words, (var, 0), 7 words of
    padding; then C rows of W; then the taps); C N > 16 (the wide kernel) starts the block and its W section on lines; smaller shapes are
    packed back to back and padded to 16 bytes."""
    CN = C * N
    npk = CN * (CN + 1) // 2
    if CN == 16:
        return dict(w0=144, x0=144 + 16 * C, var_f=2 * 136, floats=2 * (144 + 16 * C + 16), npk=npk)
    if CN > 16:                                             # the wide kernel: the block and its W section on 128-byte lines
        w0 = (npk + 15) & ~15
        x0 = w0 + C * CN
        return dict(w0=w0, x0=x0, var_f=2 * (x0 + CN), floats=(2 * (x0 + CN) + 1 + 31) & ~31, npk=npk)
    return dict(w0=npk, x0=npk + C * CN, var_f=2 * (npk + C * CN + CN), floats=(2 * (npk + C * CN + CN) + 1 + 3) & ~3, npk=npk)


class NsOmlsaMulti(_Base):
    """Multichannel (TBRR) OMLSA noise estimate and gain — noise_estimation/omlsa_multi.py:27-156."""

    def __init__(self, nfft=256, M=4, cal_weights=False, batch=1, device=-1):
        self.nfft, self.half_bin, self.M, self.cal_weights, self.batch = nfft, int(nfft / 2 + 1), M, cal_weights, int(batch)
        self._eng = BatchEngine(L.ALGO_OMLSA, M, nfft, batch=batch, device=device)
        self._o = 5 * M + 1 + (M - 1)
        self._first = True

    def estimation(self, y, u):
        """y [half_bin] beam power, u [half_bin, M-1] reference powers -> lambda_d (None on the first call)."""
        y = self._add_batch(y, 1)
        u = self._add_batch(u, 2)
        assert y.shape[-1] == self.half_bin
        lam, G, p = self._eng.omlsa_estimate(y[:, None, :], u[:, None, :, :])
        if self._first:                                                     # omlsa_multi.py:87-93 returns nothing
            self._first = False
            return None
        return self._sq(lam[:, 0, :].astype(np.float64))

    def estimation_frames(self, y, u):
        """T successive estimation() calls as one native call: y [B, T, half_bin], u [B, T, half_bin, M-1] powers (T may be 0)."""
        y = np.asarray(y, dtype=np.float32)
        u = np.asarray(u, dtype=np.float32)
        assert y.ndim == 3 and u.ndim == 4 and y.shape[-1] == self.half_bin and u.shape[-1] == self.M - 1
        if y.shape[1] == 0:
            return
        self._eng.omlsa_estimate(y, u)
        self._first = False

    def _row(self, f):
        return self._sq(self._eng.op_state()[:, f, :].astype(np.float64))

    lambda_d = property(lambda s: s._row(s._o + 0))
    gamma = property(lambda s: s._row(s._o + 1))
    G_H1 = property(lambda s: s._row(s._o + 2))
    p = property(lambda s: s._row(s._o + 4))
    xi_hat = property(lambda s: s._row(s._o + 5))
    q_hat = property(lambda s: s._row(s._o + 6))

    @property
    def G(self):
        g = self._row(self._o + 3)
        return g if self.cal_weights else np.ones_like(g)                   # :152-154 only with cal_weights


class _SubbandBase(_Base):
    def _W(self, N, C):
        st = self._eng.op_state()                                           # rows: W [N][C] (re, im) first
        w = st[:, : 2 * N * C, :].reshape(self.batch, N, C, 2, self.half_band)
        return (w[:, :, :, 0, :] + 1j * w[:, :, :, 1, :]).astype(np.complex128)   # [B, N, C, K]

    def _td(self, a):
        return 'float' in str(np.asarray(a).dtype)


class SubbandLMS(_SubbandBase):
    """Per-band N-tap (N)LMS — adaptivefilter/SubbandLMS.py:12-84 (SubbandAF.py:12-111)."""

    def __init__(self, filter_len=2, num_bands=512, mu=0.1, normalization=True, alpha=0.9, m=2, hop_length=None,
                 input_td=False, batch=1, device=-1):
        self.filter_len, self.half_band, self.batch = filter_len, int(num_bands / 2) + 1, int(batch)
        self.hop_length = int(num_bands / 2) if hop_length is None else hop_length
        self.num_bands = num_bands
        self._eng = BatchEngine(L.ALGO_SUBLMS, 1, num_bands, batch=batch, device=device, filter_len=filter_len,
                                no_norm=not normalization, filt_mu=mu, filt_alpha=alpha)
        self._tx = self._td_ = None
        self._device = device

    def _transforms(self):
        if self._tx is None:
            self._tx = Transform(n_fft=self.num_bands, hop_length=self.hop_length, batch=self.batch, device=self._device)
            self._td_ = Transform(n_fft=self.num_bands, hop_length=self.hop_length, batch=self.batch, device=self._device)
        return self._tx, self._td_

    def update(self, x_n, d_n, alpha=1e-4, p=None):
        """x_n, d_n: [half_band] complex (or [hop] float -> analysed/synthesised internally); p scalar/[half_band]/None."""
        x_n, d_n = np.asarray(x_n), np.asarray(d_n)
        assert x_n.shape == d_n.shape, 'x_n and d_n must be same shape of [samples, ]'
        td = self._td(x_n) and self._td(d_n)
        if td:                                                              # SubbandAF.py:54-57
            tx, tdd = self._transforms()
            x_n = np.squeeze(tx.analysis(x_n))
            d_n = np.squeeze(tdd.analysis(d_n))
        x = self._add_batch(x_n, 1)
        d = self._add_batch(d_n, 1)
        pp = None
        if p is not None:
            pp = np.broadcast_to(np.asarray(p, dtype=np.float32).reshape(self.batch, -1) if np.ndim(p) else
                                 np.full((self.batch, 1), p, dtype=np.float32), (self.batch, self.half_band))
            pp = pp[:, None, :]
        err = self._eng.sublms_update(x[:, None, :, None], d[:, None, :], pp)[:, 0, :]
        W = self.W
        if td:
            return tdd.synthesis(self._sq(err)), W
        return self._sq(err.astype(np.complex128)), W

    @property
    def W(self):
        return self._sq(np.transpose(self._W(self.filter_len, 1)[:, :, 0, :], (0, 2, 1)))   # [half_band, filter_len]


class SubbandLmsMc(_SubbandBase):
    """Multichannel per-band (N)LMS — adaptivefilter/SubbandLmsMc.py:13-191."""

    def __init__(self, filter_len=2, num_bands=512, channel=1, mu=0.1, normalization=True, alpha=0.9, m=2,
                 hop_length=None, input_td=False, batch=1, device=-1):
        self.filter_len, self.half_band, self.M, self.batch = filter_len, int(num_bands / 2) + 1, channel, int(batch)
        self.hop_length = int(num_bands / 2) if hop_length is None else hop_length
        self.num_bands = num_bands
        self._eng = BatchEngine(L.ALGO_SUBLMS, channel, num_bands, batch=batch, device=device, filter_len=filter_len,
                                no_norm=not normalization, filt_mu=mu, filt_alpha=alpha)
        self._tx = self._td_ = None
        self._device = device

    def update(self, x_n, d_n, alpha=1e-4, p=None):
        """x_n [half_band, 1, channel] complex (or [samples, channel] float), d_n [half_band] (or [samples]);
        p [half_band, 1] or None."""
        x_n, d_n = np.asarray(x_n), np.asarray(d_n)
        td = self._td(x_n) and self._td(d_n)
        if td:                                                              # SubbandLmsMc.py:86-92
            if self._tx is None:
                self._tx = Transform(n_fft=self.num_bands, hop_length=self.hop_length, channel=self.M, batch=self.batch,
                                     device=self._device)
                self._td_ = Transform(n_fft=self.num_bands, hop_length=self.hop_length, batch=self.batch, device=self._device)
            x_n = self._tx.analysis(x_n)
            d_n = np.squeeze(self._td_.analysis(d_n))
        x = self._add_batch(x_n, 3)[:, :, 0, :]                             # [B, K, C]  (SubbandLmsMc.py:93)
        d = self._add_batch(d_n, 1)
        pp = None
        if p is not None:
            pp = np.asarray(p, dtype=np.float32).reshape(self.batch, 1, self.half_band)
        err = self._eng.sublms_update(x[:, None, :, :], d[:, None, :], pp)[:, 0, :]
        if td:
            return self._td_.synthesis(self._sq(err)), self.W
        return self._sq(err.astype(np.complex128)), self.W

    @property
    def W(self):
        return self._sq(np.transpose(self._W(self.filter_len, self.M), (0, 3, 1, 2)))   # [half_band, filter_len, channel]


class SubbandRLS(_SubbandBase):
    """Per-band N-tap RLS — adaptivefilter/SubbandRLS.py:12-71."""

    def __init__(self, filter_len=2, num_bands=512, forgetting_factor=0.998, mu=0.5, normalization=True, alpha=0.9, m=2,
                 hop_length=None, input_td=False, batch=1, device=-1):
        self.filter_len, self.half_band, self.batch = filter_len, int(num_bands / 2) + 1, int(batch)
        self.hop_length = int(num_bands / 2) if hop_length is None else hop_length
        self.num_bands = num_bands
        self.forgetting_factor = forgetting_factor
        self._eng = BatchEngine(L.ALGO_SUBRLS, 1, num_bands, batch=batch, device=device, filter_len=filter_len,
                                filt_mu=mu, rls_lambda=forgetting_factor)
        self._tx = self._td_ = None
        self._device = device

    def update(self, x_n, d_n, alpha=1e-4, p=None):
        x_n, d_n = np.asarray(x_n), np.asarray(d_n)
        td = self._td(x_n) and self._td(d_n)
        if td:
            if self._tx is None:
                self._tx = Transform(n_fft=self.num_bands, hop_length=self.hop_length, batch=self.batch, device=self._device)
                self._td_ = Transform(n_fft=self.num_bands, hop_length=self.hop_length, batch=self.batch, device=self._device)
            x_n = np.squeeze(self._tx.analysis(x_n))
            d_n = np.squeeze(self._td_.analysis(d_n))
        x = self._add_batch(x_n, 1)
        d = self._add_batch(d_n, 1)
        err = self._eng.subrls_update(x[:, None, :], d[:, None, :])[:, 0, :]
        if td:
            return self._td_.synthesis(self._sq(err)), self.W
        return self._sq(err.astype(np.complex128)), self.W

    @property
    def W(self):
        return self._sq(np.transpose(self._W(self.filter_len, 1)[:, :, 0, :], (0, 2, 1)))

    @property
    def P(self):
        N = self.filter_len
        st = self._eng.op_state()[:, 4 * N: 4 * N + 2 * N * N, :].reshape(self.batch, N, N, 2, self.half_band)
        P = (st[:, :, :, 0, :] + 1j * st[:, :, :, 1, :]).astype(np.complex128)
        return self._sq(np.transpose(P, (0, 3, 1, 2)))                      # [half_band, N, N]


class _LiveArray(object):
    """What the reference hands back as `self.W`: an alias of the object's live state, not a snapshot (awpe.py:192 returns the array the
    next update() mutates in place).  Here the state lives in HBM, so the alias reads it when it is LOOKED AT (np.asarray, indexing, any
    ndarray attribute) — update() itself moves nothing device-to-host (the prediction filters of a 4 x 20 Wpe are 3.8 MB per utterance)."""

    def __init__(self, fetch):
        self._fetch = fetch

    def __array__(self, dtype=None, copy=None):
        a = self._fetch()
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, i):
        return self._fetch()[i]

    def __len__(self):
        return len(self._fetch())

    def __getattr__(self, name):
        return getattr(self._fetch(), name)


class Wpe(_SubbandBase):
    """RLS-based online WPE dereverberation — dereverberation/awpe.py:28-192.

    The reference's Wpe does not run at HEAD (undefined check_input_data, awpe.py:150) and sits on the
    Nyquist filterbank; this class implements the same equations on the STFT (Transform) grid with
    check_input_data(xd, x) := (analysis(xd), analysis(x)) — the semantics the golden vectors g10 / g20 pin
    (tests/golden/make_golden.py R6, R7).  update() is ONE native call (DS_ALGO_WPE_TD: analysis -> delay line of `delay`
    frames on the device -> RLS-WPE -> synthesis of channel 0) and returns the dereverberated channel 0 in the time domain.
    channels * filter_len <= 80 (the maintained use, example/wpe.ipynb cell 2: channels=4, filter_len=20, num_bands=256,
    hop_length=64); hop_length = num_bands / 2 or num_bands / 4."""

    def __init__(self, channels=2, filter_len=2, num_bands=512, forgetting_factor=0.998, delay=4, mu=0.5,
                 normalization=True, alpha=0.9, m=2, hop_length=None, input_td=False, batch=1, device=-1, precision="single"):
        """precision="double" (not in the reference's signature): the RLS recursion — P, W, the tap buffer, var — in double like the
        reference's complex128 arrays (awpe.py:60-71) instead of fp32 (DS_PARAM_WPE_FP64, csrc/ds_wpe64.hpp): several times slower, for
        stationary strongly reverberant streams where the fp32 recursion's eps x cond(P) shows (5e-4 of the output at cond(P) = 2.6e5)."""
        self.channels, self.filter_len, self.half_band, self.batch = channels, filter_len, int(num_bands / 2) + 1, int(batch)
        self.hop_length = int(num_bands / 2) if hop_length is None else hop_length
        self.D = delay
        self.forgetting_factor = forgetting_factor
        if precision not in ("single", "double"):
            raise ValueError("precision must be 'single' or 'double'")
        self.precision = precision
        self._eng = BatchEngine(L.ALGO_WPE_TD, channels, num_bands, hop=self.hop_length, batch=batch, device=device,
                                filter_len=filter_len, rls_lambda=forgetting_factor)
        self._eng.set_wpe_delay(delay)
        if precision == "double":
            self._eng.set_param_i(L.PARAM_WPE_FP64, 1)

    def update(self, x_n, alpha=1e-4, p=None):
        """x_n [hop, channels] float (or any multiple of hop samples: successive hops, bit for bit the hop-by-hop result)
        -> (dereverberated channel 0 [samples], W [half_band, channels, channels*filter_len])."""
        x = self._add_batch(x_n, 2)
        if x.shape[1] % self.hop_length != 0 or x.shape[2] != self.channels:
            raise ValueError("Wpe.update takes [k * hop (%d), channels (%d)] samples per call" % (self.hop_length, self.channels))
        y = self._eng.process(x, L.LAYOUT_SAMPLES_CHANNELS)
        return self._sq(y.astype(np.float64)), _LiveArray(lambda: self.W)

    def _blocks(self):
        """per-bin state blocks [B, K, words] complex (wpe_block_layout(): the upper triangle of P by columns, P[i][q] (i <= q) at
        q (q + 1) / 2 + i; W[c, :] from word w0; the taps from word x0)."""
        lay = wpe_block_layout(self.channels, self.filter_len)
        SB = lay["floats"]
        raw = self._eng.stage_state_raw(1).reshape(self.batch, -1)[:, : self.half_band * SB].reshape(self.batch, self.half_band, SB)
        return raw[:, :, : SB & ~1].copy().view(np.complex64)

    def _blocks64(self):
        """the double-precision state (DS_FIELD_WPE_STATE64): (P [B, K, CN, CN], W [B, K, C, CN]) complex128"""
        C, CN, K = self.channels, self.channels * self.filter_len, self.half_band
        sb = 2 * CN * CN + 2 * C * CN + 2 * CN + 2
        raw = self._eng.stage_state_f64(1, L.FIELD_WPE_STATE64).reshape(self.batch, K, sb)
        P = raw[:, :, : 2 * CN * CN].copy().view(np.complex128).reshape(self.batch, K, CN, CN)
        W = raw[:, :, 2 * CN * CN: 2 * CN * CN + 2 * C * CN].copy().view(np.complex128).reshape(self.batch, K, C, CN)
        return P, W

    @property
    def W(self):
        if self.precision == "double":
            return self._sq(self._blocks64()[1])
        C, CN = self.channels, self.channels * self.filter_len
        w0 = wpe_block_layout(C, self.filter_len)["w0"]
        w = self._blocks()[:, :, w0:w0 + C * CN].reshape(self.batch, self.half_band, C, CN)
        return self._sq(w.astype(np.complex128))                                       # [half_band, C, C*N]

    @property
    def P(self):
        if self.precision == "double":
            return self._sq(self._blocks64()[0])
        CN = self.channels * self.filter_len
        blk = self._blocks()
        P = np.zeros((self.batch, self.half_band, CN, CN), dtype=np.complex128)
        for q in range(CN):
            for i in range(q + 1):
                P[:, :, i, q] = blk[:, :, q * (q + 1) // 2 + i]
                P[:, :, q, i] = np.conj(blk[:, :, q * (q + 1) // 2 + i])
        return self._sq(P)                                                             # [half_band, CN, CN]


class BaseFilter(_Base):
    """Sample-wise time-domain NLMS — adaptivefilter/BaseFilter.py:25-110 (filter_len <= 1024 on the GPU)."""

    def __init__(self, filter_len=1024, mu=0.1, normalization=True, batch=1, device=-1):
        self.filter_len, self.mu, self.norm, self.batch = filter_len, mu, normalization, int(batch)
        self._eng = BatchEngine(L.ALGO_TDNLMS, 1, 512, batch=batch, device=device, filter_len=filter_len, filt_mu=mu,
                                no_norm=not normalization)

    def update(self, x_n, d_n, eps=1e-4, p=1.0):
        """one sample in -> (err, w [filter_len, 1])."""
        x = np.asarray(x_n, dtype=np.float32).reshape(self.batch, 1)
        d = np.asarray(d_n, dtype=np.float32).reshape(self.batch, 1)
        err = self._eng.tdfilter_update(x, d, p=p)[:, 0]
        return self._sq(err.astype(np.float64)), self.w

    def filter(self, data, data_d):
        """run the whole signal (n successive update() calls in one kernel launch) -> err [n]."""
        x = self._add_batch(np.asarray(data, dtype=np.float32), 1)
        d = self._add_batch(np.asarray(data_d, dtype=np.float32), 1)
        return self._sq(self._eng.tdfilter_update(x, d).astype(np.float64))

    @property
    def w(self):
        return self._sq(self._eng.tdfilter_weights().astype(np.float64)[:, :, None])


class Rls(BaseFilter):
    """Sample-wise time-domain RLS — adaptivefilter/RLS.py:14-42 (filter_len <= 256: P lives in LDS up to 64 taps, in device memory beyond)."""

    def __init__(self, filter_len=1024, mu=0.5, forgetting_factor=0.9998, delta=1e-3, normalization=True, batch=1, device=-1):
        if delta != 1e-3:
            raise NotImplementedError("delta is fixed to the reference default 1e-3")
        self.filter_len, self.mu, self.norm, self.batch = filter_len, mu, normalization, int(batch)
        self.forgetting_factor = forgetting_factor
        self._eng = BatchEngine(L.ALGO_TDRLS, 1, 512, batch=batch, device=device, filter_len=filter_len, filt_mu=mu,
                                rls_lambda=forgetting_factor)

    def update(self, x_n, d_n, alpha=1e-4):
        return BaseFilter.update(self, x_n, d_n)


class FastFreqLms(_Base):
    """Overlap-save frequency-domain block LMS — adaptivefilter/FastFreqLms.py:48-245.
    One `update` = one block of `filter_len` samples through the ds_fdaf kernel (n_fft = 2 * filter_len,
    filter_len in {64, 128, 256, 512}, n_channels <= 8); `filter` runs a whole signal in one launch."""

    _KIND = L.FDAF_PLAIN

    def __init__(self, filter_len=128, hop_len=None, win_len=None, mu=0.01, constrain=True, n_channels=1, alpha=0.9,
                 non_causal=False, two_path=False, batch=1, device=-1, weight_norm=False):
        if two_path and self._KIND != L.FDAF_PLAIN:
            raise NotImplementedError("two_path belongs to the plain FastFreqLms (the GSC filters of the reference never set it)")
        self.two_path = bool(two_path)
        if (hop_len is not None and hop_len != filter_len) or (win_len is not None and win_len != 2 * filter_len):
            raise NotImplementedError("only the default framing hop_len = filter_len, win_len = 2 * filter_len is built")
        self.filter_len, self.hop_len, self.win_len, self.mu = filter_len, filter_len, 2 * filter_len, mu
        self.n_channels, self.alpha, self.constrain, self.non_causal = n_channels, alpha, constrain, non_causal
        self.n_fft = 2 ** (int(np.log2(self.hop_len + filter_len - 1)) + 1)                  # :70-71
        if self.n_fft != 2 * filter_len:
            raise NotImplementedError("filter_len must be a power of two (n_fft = 2 * filter_len); got %d" % filter_len)
        self.overlap = self.win_len - self.hop_len
        self.batch = int(batch)
        self._eng = BatchEngine(L.ALGO_FDAF, n_channels, self.n_fft, batch=batch, device=device, filt_mu=mu, filt_alpha=alpha)
        self._eng.set_fdaf(self._KIND, constrain=constrain, non_causal=non_causal, weight_norm=weight_norm, two_path=two_path)
        self._w = np.zeros((self.batch, filter_len, n_channels))

    def update(self, x_n_vec, d_n_vec, update=True, p=1.0, fir_truncate=None, filter_p=False):
        """x [hop] or [hop, C]; d [hop] or [hop, 1]; p scalar or [K, 1] -> (e [hop, 1], w [filter_len, C])."""
        x = np.asarray(x_n_vec, dtype=np.float32)
        d = np.asarray(d_n_vec, dtype=np.float32)
        if self.batch == 1:
            x = x.reshape(1, self.hop_len, self.n_channels)
            d = d.reshape(1, self.hop_len)
        else:
            x = x.reshape(self.batch, self.hop_len, self.n_channels)
            d = d.reshape(self.batch, self.hop_len)
        K = self.n_fft // 2 + 1
        if not update:
            pp = np.zeros((self.batch, 1), dtype=np.float32)                                  # W + 0 * grad
        elif np.ndim(p) == 0:
            pp = None if float(p) == 1.0 else np.full((self.batch, 1), float(p), dtype=np.float32)
        else:
            pp = np.asarray(p, dtype=np.float32).reshape(self.batch, 1, K)
        e, w = self._eng.fdaf_update(x, d, p=pp, fir_truncate=fir_truncate)
        self._w = w.astype(np.float64)
        return self._sq(e.astype(np.float64)[:, :, None]), self.w

    def filter(self, x, d, p=None, fir_truncate=None):
        """whole signal: x [n, C] (n a multiple of filter_len), d [n], p None | [T] | [T, K] -> e [n]."""
        x = np.asarray(x, dtype=np.float32)
        if self.n_channels == 1 and x.ndim == (1 if self.batch == 1 else 2):
            x = x[..., None]
        x = self._add_batch(x, 2)
        d = self._add_batch(np.asarray(d, dtype=np.float32), 1)
        if p is not None:
            p = np.asarray(p, dtype=np.float32)
            p = p[None] if self.batch == 1 else p
        e, w = self._eng.fdaf_update(x, d, p=p, fir_truncate=fir_truncate)
        self._w = w.astype(np.float64)
        return self._sq(e.astype(np.float64))

    @property
    def w(self):
        return self._sq(self._w)

    @property
    def W(self):
        W, _ = self._eng.fdaf_state()
        return self._sq(np.swapaxes(W, 1, 2).astype(np.complex128))                          # [K, C]

    @property
    def P(self):
        _, P = self._eng.fdaf_state()
        return self._sq(P.astype(np.float64)[:, :, None])


    @property
    def foreground(self):
        """foreground filter [n_fft / 2 + 1, n_channels] of a two-path filter (FastFreqLms.py:95-96)."""
        F = self._eng.fdaf_foreground().astype(np.complex128)                 # [B, C, K]
        F = np.swapaxes(F, 1, 2)
        return F[0] if self.batch == 1 else F


class AdaptiveBlockingMatrixFilter(FastFreqLms):
    """Coefficient-clamped FDAF of the adaptive blocking matrix — beamformer/gsc_bm.py:22-122."""
    _KIND = L.FDAF_BM


class AdaptiveInterferenceCancellation(FastFreqLms):
    """Norm-limited FDAF of the interference canceller — beamformer/gsc_aic.py:25-108."""
    _KIND = L.FDAF_AIC

    def __init__(self, filter_len=128, hop_len=None, win_len=None, mu=0.01, constrain=True, weight_norm=False, n_channels=1,
                 alpha=0.9, non_causal=False, two_path=False, batch=1, device=-1):
        FastFreqLms.__init__(self, filter_len=filter_len, hop_len=hop_len, win_len=win_len, mu=mu, constrain=constrain,
                             n_channels=n_channels, alpha=alpha, non_causal=non_causal, two_path=two_path, batch=batch,
                             device=device, weight_norm=weight_norm)
        self.weight_norm = weight_norm
