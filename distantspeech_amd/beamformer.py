"""Enhancement objects with the reference's names, constructor arguments and process() contracts,
backed by the native batched engine (libdsenh.so HIP kernels).

  beamformer         base: geometry set-up, DS / SD weights      (beamformer/beamformer.py:218-373)
  FixedBeamformer    process(x[L, M], angle) -> y[L]              (beamformer/fixedbeamformer.py:96-207)
  adaptivebeamfomer  process(x[M, L], angle, method) -> dict      (beamformer/adaptivebeamformer.py:10-128)
  GSC                process(x[M, L], angle, method) -> dict      (beamformer/GSC.py:26-294)

Every class takes an extra ``batch`` argument: with batch=B, process() takes [B, ...] arrays and
B independent utterances run as one kernel launch; with the default batch=1 the shapes are
exactly the reference's.  A call with T hops equals T successive one-hop reference calls."""
import numpy as np

from . import _lib as L
from .engine import BatchEngine
from .mic_array import MicArray, compute_tau, gen_noise_msc


def compute_mvdr_weight(steer_vector, Rvv_inv):
    """w = R^-1 a / (a^H R^-1 a) per bin — beamformer/beamformer.py:133-155 (host set-up)."""
    num = Rvv_inv @ steer_vector[..., None]
    w = num / (steer_vector[:, None, :].conj() @ num)
    return w.squeeze()


class _McraView(object):
    """Read-only view of the engine's MCRA state with the reference attribute names
    (noise_estimation/NoiseEstimationBase.py:11-31, mcra.py:20-25)."""

    def __init__(self, owner):
        self._o = owner

    def _f(self, field):
        return self._o._squeeze(self._o._eng.get_field(field).astype(np.float64))

    S = property(lambda s: s._f(L.FIELD_MCRA_S))
    Smin = property(lambda s: s._f(L.FIELD_MCRA_SMIN))
    Stmp = property(lambda s: s._f(L.FIELD_MCRA_STMP))
    p = property(lambda s: s._f(L.FIELD_MCRA_P))
    lambda_d = property(lambda s: s._f(L.FIELD_MCRA_LAMBDA_D))

    @property
    def frm_cnt(self):
        return self._o._squeeze(self._o._eng.get_field(L.FIELD_COUNTERS)[:, 0])

    @property
    def L(self):
        return self._o._mcra_L

    @L.setter
    def L(self, value):          # users poke `obj.mcra.L = 10` in the reference's notebooks
        self._o._mcra_L = int(value)
        self._o._eng.set_mcra_L(int(value))


class beamformer(object):
    """beamformer base class — beamformer/beamformer.py:218-373."""

    _ALGO = None

    def __init__(self, mic: MicArray, frame_len=256, hop=None, nfft=None, batch=1, device=-1):
        self.MicArray = mic
        self.M = mic.M
        self.frameLen = frame_len
        self.hop = int(frame_len // 2) if hop is None else int(hop)
        self.overlap = frame_len - self.hop
        self.nfft = int(frame_len) if nfft is None else int(nfft)
        self.c, self.r, self.fs = mic.c, mic.r, mic.fs
        self.half_bin = round(self.nfft / 2 + 1)
        self.freq_bin = np.linspace(0, self.half_bin - 1, self.half_bin)
        self.omega = 2 * np.pi * self.freq_bin * self.fs / self.nfft
        self.W = np.zeros((self.half_bin, self.M), dtype=complex)
        self.Fvv = gen_noise_msc(mic=mic, nfft=self.nfft)
        self.batch = int(batch)
        self._device = device
        self._eng = None
        # process() returns float64 samples like the reference (transform.py:479 scales a float32 buffer by a Python float): widened on the
        # device (ds_process_f64; exact).  `out_dtype = np.float32` hands back the kernel's own samples and halves the download
        self.out_dtype = np.float64

    # -- geometry / fixed weights (host set-up) ----------------------------------------------------
    def compute_steering_vector_from_doa(self, look_angle=(0, 0)):
        """a0 [bins, M] for a look direction in degrees — beamformer.py:267-289."""
        look_angle_rad = np.array(look_angle) / 180 * np.pi
        tau0 = compute_tau(self.MicArray, look_angle_rad)
        a0 = np.zeros((self.half_bin, self.M), dtype=complex)
        kmax = min(self.half_bin, self.MicArray.half_bin)      # the reference loops mic_array.half_bin (:286)
        a0[:kmax, :] = np.exp(-1j * self.omega[:kmax, None] * tau0[None, :, 0])
        return a0

    def compute_weights(self, look_angle=[90, 0], weightType="DS", diag_value=1e-3):
        """DS / SD (superdirective) weights [bins, M] — beamformer.py:338-373."""
        a0 = self.compute_steering_vector_from_doa(look_angle=look_angle)
        if weightType == 'DS':
            return a0 / self.M
        if weightType == 'SD':
            diag_bin = np.broadcast_to(np.eye(self.M) * diag_value, (self.half_bin, self.M, self.M))
            return compute_mvdr_weight(a0, np.linalg.inv(self.Fvv + diag_bin))
        raise ValueError("Unknown beamformer weights: %s" % weightType)

    # -- helpers ------------------------------------------------------------------------------------
    def _make_engine(self, **kw):
        self._eng = BatchEngine(self._ALGO, self.M, self.nfft, self.hop, batch=self.batch, device=self._device, **kw)

    def _squeeze(self, a):
        return a[0] if self.batch == 1 else a

    def _as_batch(self, x, ndim_single):
        x = np.asarray(x)
        if x.ndim == ndim_single:
            if self.batch != 1:
                raise ValueError("this object was built with batch=%d; pass [B, ...] arrays" % self.batch)
            x = x[None]
        return x

    def reset(self):
        self._eng.reset()


class FixedBeamformer(beamformer):
    """Delay-and-sum / superdirective beamformer — beamformer/fixedbeamformer.py:96-207.

    The reference's own compute_weights builds the diffuse coherence with nfft=256 regardless of
    the object's nfft (fixedbeamformer.py:140), which only works at nfft=256; the base-class
    semantics (Fvv at the object's nfft, beamformer.py:262,369-371) are used instead."""

    _ALGO = L.ALGO_FIXED

    def __init__(self, MicArray, frameLen=256, hop=None, nfft=None, c=343, fs=16000, r=0.032, weightType="SD",
                 batch=1, device=-1):
        beamformer.__init__(self, MicArray, frame_len=frameLen, hop=hop, nfft=nfft, batch=batch, device=device)
        self.angle = [197, 0]
        self.weightType = weightType
        self.AlgorithmList = ["src", "DS", "MVDR"]
        self.AlgorithmIndex = 0
        self._make_engine()
        self.W = self.compute_weights(look_angle=self.angle, weightType=weightType)
        self._eng.set_steering(self.W)

    def process(self, x, angle=(0, 0)):
        """x [samples, channel] (or [B, samples, channel]) -> [samples] (or [B, samples])."""
        x = self._as_batch(x, 2)
        assert x.shape[2] >= 2
        angle = list(angle)
        if angle != self.angle:                                   # fixedbeamformer.py:183-188
            self.angle = angle
            self.W = self.compute_weights(look_angle=angle, weightType=self.weightType)
            self._eng.set_steering(self.W)
        if x.shape[1] % self.hop != 0:
            raise ValueError("samples (%d) must be a multiple of hop (%d)" % (x.shape[1], self.hop))
        y = self._eng.process(x, L.LAYOUT_SAMPLES_CHANNELS, out_dtype=self.out_dtype)
        return self._squeeze(y).astype(self.out_dtype, copy=False)


class _AdaptiveBase(beamformer):
    """shared plumbing of adaptivebeamfomer and GSC: x [M, L] in, dict out, circular-array tao."""

    def _tao(self, angle):
        return -1 * self.r * np.cos(angle[1]) * np.cos(angle[0] - self.gamma) / self.c

    def _update_steering(self, angle):
        angle = np.asarray(angle, dtype=float)
        if self._steer_angle is None or not np.array_equal(angle, self._steer_angle):
            tao = self._tao(angle)
            self._a = np.exp(-1j * self.omega[:, None] * tao[None, :])      # [K, M]
            self._eng.set_steering(self._a)
            self._steer_angle = angle.copy()
            self.angle = angle

    def beampattern(self, omega, H):
        """beamformer.beampattern (beamformer.py:536-553): 10 log10 |H[:, k]^H a(az, k)| over azimuth 0 .. 359 degrees of a circular array of radius
        0.032 (hard-wired there) at elevation 0: [360, half_bin] (or [B, 360, half_bin] for a batched H [B, M, half_bin])."""
        H = np.asarray(H)
        r = 0.032
        az = np.arange(0, 360, 1) * np.pi / 180
        tao = -1 * r * np.cos(0) * np.cos(az[:, None] - np.asarray(self.gamma)[None, :]) / self.c            # [360, M]
        a = np.exp(-1j * np.asarray(omega)[None, :, None] * tao[:, None, :])                             # [360, K, M]
        with np.errstate(divide="ignore"):
            return 10 * np.log10(np.abs(np.einsum("...mk,zkm->...zk", H.conj(), a)))

    def _run(self, x, angle, method, retH, retWNG, retDI):
        if retWNG or retDI:
            # the reference calls undefined calcWNG / calcDI here (adaptivebeamformer.py:115-117, GSC.py:276-279)
            raise AttributeError("'%s' object has no attribute 'calcWNG'" % type(self).__name__)
        x = self._as_batch(x, 2)
        if x.shape[1] != self.M:
            raise ValueError("x must be [channels=%d, samples]" % self.M)
        if x.shape[2] % self.hop != 0:
            raise ValueError("samples (%d) must be a multiple of hop (%d)" % (x.shape[2], self.hop))
        self._update_steering(angle)
        if method != self.AlgorithmIndex:
            self.AlgorithmIndex = method
            self._eng.set_method(method)
        self._before_process()
        y = self._eng.process(x, L.LAYOUT_CHANNELS_SAMPLES, out_dtype=self.out_dtype)
        # adaptivebeamformer.py:124-126, GSC.py:290-292: beampattern of the weights the object holds after the call's last frame
        bp = self.beampattern(self.omega, self._weights_after_call(method)) if retH else None
        return {'data': self._squeeze(y).astype(self.out_dtype, copy=False), 'WNG': None, 'DI': None, 'beampattern': bp}

    def _before_process(self):
        pass

    def _weights_after_call(self, method):
        """`self.H` [M, half_bin] as the reference object holds it when process() returns."""
        raise NotImplementedError


class _SppView(object):
    """`spp` of an adaptivebeamfomer(postfilter="mcmcra"): the McMcra state the fused kernel carries (mc_mcra.py:68-69)."""

    def __init__(self, owner):
        self._o = owner

    Phi_yy = property(lambda s: s._o._squeeze(s._o._eng.get_field(L.FIELD_PHI_YY).astype(np.float64)))
    Phi_vv = property(lambda s: s._o._squeeze(s._o._eng.get_field(L.FIELD_PHI_VV).astype(np.float64)))
    frm_cnt = property(lambda s: s._o._squeeze(s._o._eng.get_field(L.FIELD_COUNTERS)[:, 2]))


class adaptivebeamfomer(_AdaptiveBase):
    """MCRA-gated adaptive beamformer (src / DS / MVDR / TFGSC) — beamformer/adaptivebeamformer.py:10-128.

    postfilter="mcmcra" (not in the reference's class; BASELINE's "MVDR + post-filter" workload): the McMcra speech-presence gain of the
    same input frame multiplies the beamformer output, Y = (H^H Z) spp.G — GSC.process's convention for its `spp` (GSC.py:225,286) — inside
    the same fused frame kernel (DS_ALGO_ADAPTIVE_PF).  Methods src / DS / MVDR; `self.spp.Phi_yy / Phi_vv` read the McMcra state."""

    _ALGO = L.ALGO_ADAPTIVE

    def __init__(self, mic: MicArray, frameLen=256, hop=None, nfft=None, c=343, r=0.032, fs=16000, batch=1,
                 device=-1, track_ryy=True, postfilter=None):
        beamformer.__init__(self, mic, frame_len=frameLen, hop=hop, nfft=nfft, batch=batch, device=device)
        if postfilter not in (None, False, "mcmcra"):
            raise ValueError("postfilter must be None or 'mcmcra', got %r" % (postfilter,))
        self.postfilter = postfilter or None
        if self.postfilter:
            self._ALGO = L.ALGO_ADAPTIVE_PF
            track_ryy = False                  # the one-pass kernel keeps no Ryy (TFGSC is refused)
            self.spp = _SppView(self)
        self.gamma = mic.gamma
        self.angle = np.array([0, 0]) / 180 * np.pi
        self.method = 'MVDR'
        self.estPos = None             # None: the MCRA-based gate; n: Rvv follows the first n (frame, bin) slots only (adaptivebeamformer.py:30,90-93)
        self._est_pos_native = None
        self.AlgorithmList = ['src', 'DS', 'MVDR', 'TFGSC']
        self.AlgorithmIndex = 0
        self._mcra_L = 15
        self._steer_angle = None
        self._make_engine(track_ryy=track_ryy)
        self._eng.set_method(0)
        self.mcra = _McraView(self)

    def process(self, x, angle, method=2, retH=False, retWNG=False, retDI=False):
        """x [M, samples] (or [B, M, samples]); angle = (azimuth, elevation) in radians."""
        return self._run(x, angle, method, retH, retWNG, retDI)

    def _before_process(self):
        # `estPos` is an attribute users set after construction (adaptivebeamformer.py:30): handed to the library when it changed
        if self.estPos != self._est_pos_native:
            if self.estPos is not None and int(self.estPos) < self.half_bin:
                # frameCount advances once per BIN (:90-93): with estPos < half_bin the bins from estPos on never see an update, their
                # Rvv_inv stays the zero matrix of the constructor and getweights() divides 0 by 0 — the reference's output is NaN from the
                # first sample on.  Refused instead of reproduced
                raise ValueError("estPos = %r is less than half_bin = %d: the reference's output is NaN (bins that never update keep "
                                 "Rvv_inv = 0, adaptivebeamformer.py:32-33,90-112); use a multiple of half_bin for whole frames" % (self.estPos, self.half_bin))
            self._eng.set_param_i(L.PARAM_EST_POS, -1 if self.estPos is None else int(self.estPos))
            self._est_pos_native = self.estPos

    def _weights_after_call(self, method):
        a = self._a                                                                  # [K, M]
        if method == 0:                                                              # beamformer.py:320-322: a with every channel but the first zeroed
            H = np.zeros((self.M, self.half_bin), dtype=complex); H[0] = a[:, 0]
            return H if self.batch == 1 else np.broadcast_to(H, (self.batch,) + H.shape)
        if method == 1:                                                              # :323-324
            H = a.T / self.M
            return H if self.batch == 1 else np.broadcast_to(H, (self.batch,) + H.shape)
        if method == 2:                                                              # :325-326, the kernel's own solve on the resident Rvv
            return self.H_kernel
        R = self._eng.get_field(L.FIELD_RVV).astype(np.complex128)                   # :327-333 TFGSC
        temp = np.linalg.inv(R + 1e-6 * np.eye(self.M)) @ self._eng.get_field(L.FIELD_RYY).astype(np.complex128)
        tr = np.trace(temp, axis1=-2, axis2=-1)
        col0 = temp[..., :, 0].copy(); col0[..., 0] -= 1
        return self._squeeze(np.swapaxes(col0 / (tr - self.M)[..., None], 1, 2))

    # state the reference exposes as attributes ---------------------------------------------------
    @property
    def Rvv(self):
        return self._squeeze(self._eng.get_field(L.FIELD_RVV).astype(np.complex128))

    @property
    def Ryy(self):
        return self._squeeze(self._eng.get_field(L.FIELD_RYY).astype(np.complex128))

    @property
    def Rvv_inv(self):
        """inv(Rvv + 1e-6 I) (adaptivebeamformer.py:103-104); derived on the host from Rvv on demand —
        the kernels solve (Rvv + dI) v = a directly and never form the inverse."""
        R = self._eng.get_field(L.FIELD_RVV).astype(np.complex128)
        return self._squeeze(np.linalg.inv(R + 1e-6 * np.eye(self.M)))

    @property
    def H(self):
        """current MVDR weights [M, half_bin] (adaptivebeamformer.py:105-112), derived from Rvv on demand."""
        Ri = np.linalg.inv(self._eng.get_field(L.FIELD_RVV).astype(np.complex128) + 1e-6 * np.eye(self.M))
        a = self._a if self._steer_angle is not None else np.ones((self.half_bin, self.M), dtype=complex)
        num = Ri @ a[None, :, :, None]
        w = (num / (a[None, :, None, :].conj() @ num))[..., 0]          # [B, K, M]
        return self._squeeze(np.swapaxes(w, 1, 2))


    @property
    def H_kernel(self):
        """the weights [M, half_bin] the frame kernel applies to the next frame, from the kernel's own fused Cholesky solve on the resident
        Rvv (ds_get_state DS_FIELD_H: a read-only probe running mvdr_output() with unit frames) — what `H` derives on the host with NumPy's
        inverse; methods src / DS / MVDR."""
        w = self._eng.get_field(L.FIELD_H)
        return self._squeeze(np.swapaxes(w.astype(np.complex128), 1, 2))


class GSC(_AdaptiveBase):
    """Frequency-domain GSC with SPP-controlled LMS canceller and McMcra gain — beamformer/GSC.py:26-294."""

    _ALGO = L.ALGO_GSC

    def __init__(self, mic_array: MicArray, frameLen=256, angle=[197, 0], batch=1, device=-1, track_omlsa_multi=False):
        """track_omlsa_multi: also keep `self.omlsa_multi` (NsOmlsaMulti, GSC.py:78) up to date — the reference runs its estimation() on the
        canceller output and the blocking-matrix outputs of every frame (GSC.py:281-283) and uses nothing of it, so it is off by default:
        when on, the frame kernel writes those powers out per frame and bin (DS_PARAM_REF_POWERS) and an NsOmlsaMulti operator consumes
        them after every process() call."""
        beamformer.__init__(self, mic_array, frame_len=frameLen, batch=batch, device=device)
        self.mic_array = mic_array
        self.angle = np.array(angle) / 180 * np.pi if isinstance(angle, list) else angle
        self.gamma = mic_array.gamma
        self.AlgorithmList = ['src', 'DS', 'MVDR', 'TFGSC']
        self.AlgorithmIndex = 0
        self._steer_angle = None
        self._make_engine()
        self._eng.set_method(0)
        self._track_omlsa_multi = bool(track_omlsa_multi)
        if self._track_omlsa_multi:
            from .ops import NsOmlsaMulti
            self.omlsa_multi = NsOmlsaMulti(nfft=self.nfft, M=self.M, cal_weights=True, batch=batch, device=device)   # GSC.py:78
            self._eng.set_param_i(L.PARAM_REF_POWERS, 1)

    def process(self, x, angle, method=2, retH=False, retWNG=False, retDI=False):
        """x [M, samples] (or [B, M, samples]); angle in radians; method 0 passes channel 0 through."""
        out = self._run(x, angle, method, retH, retWNG, retDI)
        if self._track_omlsa_multi and method != 0 and np.shape(x)[-1] >= self.hop:      # GSC.py:242-243: method 0 leaves the frame before :281; an empty block ran no frame
            pw = self._eng.get_field(L.FIELD_REF_POWERS)                    # [B, T, K, M]: |Y|^2, |U_1|^2 .. |U_{M-1}|^2
            self.omlsa_multi.estimation_frames(pw[..., 0], pw[..., 1:])
        return out

    def _weights_after_call(self, method):
        H = np.ones((self.M, self.half_bin), dtype=complex) / self.M                 # GSC.py:50: set in the constructor and never written again
        return H if self.batch == 1 else np.broadcast_to(H, (self.batch,) + H.shape)

    @property
    def G(self):
        """adaptive canceller weights [M-1, half_bin] (GSC.py:72)."""
        g = self._eng.get_field(L.FIELD_G_AIC).astype(np.complex128)
        return self._squeeze(np.swapaxes(g, 1, 2))

    @property
    def Phi_yy(self):
        return self._squeeze(self._eng.get_field(L.FIELD_PHI_YY).astype(np.float64))

    @property
    def Phi_vv(self):
        return self._squeeze(self._eng.get_field(L.FIELD_PHI_VV).astype(np.float64))
