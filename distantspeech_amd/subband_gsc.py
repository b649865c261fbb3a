"""SubbandGSC (BASELINE config-5 structure) and its front-end conditioning, composed from the GPU operators.

  fractional_delay_filter_bank   transform/multirate.py:4-51        (host set-up, once per look direction)
  FilterDcNotch16                adaptivefilter/feature.py:32-49    (GPU: ds_dcnotch)
  TimeAlignment                  beamformer/fixedbeamformer.py:51-93 (GPU: ds_firbank)
  DelaySamples                   beamformer/utils.py:241-274        (a buffer; no arithmetic)
  SubbandGSC                     beamformer/SubbandGSC.py:67-262

SubbandGSC.process runs behind one native chain handle (DS_ALGO_SUBBAND_GSC): FIR bank + channel mean, STFT, McSpp, the M adaptive
blocking filters as ONE batched subband-LMS/RLS launch, ISTFT, re-analysis, multichannel canceller, ISTFT — device-resident
between the stages; this module builds the constant tables and mirrors the reference's interface."""
import numpy as np

from . import _lib as L
from .engine import BatchEngine
from .mic_array import MicArray, compute_tau
from .ops import McSpp


def fractional_delay_filter_bank(delays):
    """windowed-sinc fractional delay filters [filter_len, chs] — transform/multirate.py:4-51."""
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


class FilterDcNotch16(object):
    """feature.py:32-49; `filter_dc_notch16(x)` returns (out, mem) like the reference."""

    def __init__(self, radius=0.9, device=-1):
        self.radius = radius
        self._eng = BatchEngine(L.ALGO_FRONTEND, 1, 512, batch=1, device=device, filt_alpha=radius)
        self.notch_mem = np.zeros((2,))

    def filter_dc_notch16(self, input):
        out = self._eng.dcnotch(np.asarray(input, dtype=np.float32)[None, None, :])[0, 0]
        self.notch_mem[:] = self._eng.get_field(L.FIELD_NOTCH_MEM)[0, 0]     # the live filter memory, as the reference returns it
        return out.astype(np.float64), self.notch_mem


class DelaySamples(object):
    """A delay line of `delay` samples fed `data_len` samples per call — the interface of beamformer/utils.py:241-274 (same constructor
    arguments, `delay(x)` takes [samples] or [samples, channels] and returns [samples, channels]).  Kept as a ring: the write position
    advances by the block length and the read position trails it by `delay`; nothing is shifted."""

    def __init__(self, data_len, delay, channel=1, dtype=np.float64):
        self.data_len, self.n_delay = int(data_len), int(delay)
        self._ring = np.zeros((self.data_len + self.n_delay, channel), dtype=dtype)
        self._w = self.n_delay                      # next write position; the oldest sample still owed sits `delay` behind it

    @property
    def buffer(self):
        """the line's content, oldest sample first (the reference object's `buffer` attribute after a call: its first `delay` rows)"""
        return np.roll(self._ring, -(self._w - self.n_delay), axis=0)

    def delay(self, x):
        x = np.asarray(x)
        if x.ndim == 1:
            x = x[:, None]
        if self.n_delay == 0:
            return x
        n, size = x.shape[0], self._ring.shape[0]
        if n > self.data_len:
            raise ValueError("DelaySamples was built for blocks of at most %d samples" % self.data_len)
        w = (self._w + np.arange(n)) % size
        self._ring[w] = x
        out = self._ring[(w - self.n_delay) % size]
        self._w = int((self._w + n) % size)
        return out


class TimeAlignment(object):
    """fractional-delay pre-steering — beamformer/fixedbeamformer.py:51-93."""

    def __init__(self, mic_array: MicArray, angle=[197, 0], frame_len=256, hop=None, nfft=None, r=0.032, fs=16000, batch=1,
                 device=-1):
        self.angle = np.array(angle) / 180 * np.pi if isinstance(angle, list) else angle
        self.M, self.batch = mic_array.M, int(batch)
        tau = compute_tau(mic_array, self.angle)
        self.tau = -(tau - np.max(tau))                                         # :67
        self.delay_filter = fractional_delay_filter_bank(np.array(self.tau)[:, 0] * mic_array.fs)   # :68-70
        self.delay_filter_len = self.delay_filter.shape[0]
        self._eng = BatchEngine(L.ALGO_FRONTEND, self.M, 512, batch=batch, device=device)
        self._eng.set_aux(self.delay_filter)

    def process(self, x):
        """x [samples, chs] (or [B, samples, chs]) -> aligned, same shape."""
        x = np.asarray(x)
        single = x.ndim == 2
        y, _ = self._eng.firbank(x[None] if single else x)
        return (y[0] if single else y).astype(np.float64)

    def process_with_mean(self, x):
        y, m = self._eng.firbank(x)
        return y, m


class SubbandGSC(object):
    """Subband GSC: time alignment -> mean fixed beamformer -> M SPP-controlled adaptive blocking filters ->
    multichannel adaptive interference canceller — beamformer/SubbandGSC.py:67-262.

    The whole of process() runs behind ONE native handle (DS_ALGO_SUBBAND_GSC, csrc/ds_api_chains.hip chain2_run): notch, FIR bank + mean
    beamformer, the five transforms, McSpp, the M blocking filters as one batched launch, the canceller — every stage a kernel on the
    handle's stream reading the previous stage's device buffer, all blocks of a call per launch.  `bm_filter="rls"` swaps the
    blocking filters for SubbandRLS(filter_len=2) — the BASELINE config-5 composition (defined by us, the
    reference never composes it)."""

    def __init__(self, mic_array: MicArray, frameLen=256, angle=[197, 0], batch=1, device=-1, bm_filter="lms"):
        self.M, self.frameLen, self.batch = mic_array.M, frameLen, int(batch)
        self._device = device
        self.MicArray = mic_array
        self.nfft, self.hop, self.half_bin = 2 * frameLen, frameLen, frameLen + 1
        self.angle = np.array(angle) / 180 * np.pi if isinstance(angle, list) else angle
        self.bm_filter = bm_filter
        self._eng = BatchEngine(L.ALGO_SUBBAND_GSC, self.M, self.nfft, self.hop, batch=batch, device=device, filter_len=2,
                                rls_lambda=0.998 if bm_filter == "rls" else 0.0)              # SubbandRLS.py:30 forgetting factor
        tau = compute_tau(mic_array, self.angle)
        self.tau = -(tau - np.max(tau))                                                        # fixedbeamformer.py:67
        self.delay_filter = fractional_delay_filter_bank(np.array(self.tau)[:, 0] * mic_array.fs)   # :68-70
        self.time_alignment = type("TimeAlignmentTables", (object,), dict(delay_filter=self.delay_filter, tau=self.tau,
                                                                              delay_filter_len=self.delay_filter.shape[0]))()   # :85
        self._eng.chain_set_aux(L.CHAIN_AUX_FIR, self.delay_filter)
        self._eng.chain_set_aux(L.CHAIN_AUX_COHERENCE, McSpp.diffuse_coherence(mic_array, self.nfft))

    def fixed_beamformer(self, x):
        return np.mean(x, axis=1, keepdims=True)

    def process(self, x, postfilter=False):
        """x [n_chs, n_samples] (or [B, n_chs, n_samples]) ->
        (output [L], fix_output [L], bm_output [L, M], p [half_bin, blocks], aligned_output [L, M])."""
        # postfilter=True: the reference's branch (SubbandGSC.py:236-249) analyses the canceller output, runs NsOmlsaMulti on it and scales a
        # spectrum Y that is then dropped (the synthesis at :249 is commented out) — nothing it computes reaches the five returned arrays or
        # any state they depend on.  Its one trace is the object's omlsa_multi: _postfilter_trace() below keeps that, behind the chain
        x = np.asarray(x)
        single = x.ndim == 2
        if single:
            if self.batch != 1:
                raise ValueError("object built with batch=%d; pass [B, n_chs, n_samples]" % self.batch)
            x = x[None]
        if x.shape[1] != self.M or x.shape[2] % self.frameLen != 0:
            raise ValueError("x must be [n_chs=%d, k * %d samples]" % (self.M, self.frameLen))
        y, fix, bm, p, al = self._eng.subband_gsc_process(x)
        if postfilter:
            self._postfilter_trace(y, bm)
        out = (y.astype(np.float64), fix.astype(np.float64), np.swapaxes(bm, 1, 2).astype(np.float64),
               np.swapaxes(p, 1, 2).astype(np.float64), np.swapaxes(al, 1, 2).astype(np.float64))
        return tuple(a[0] for a in out) if single else out

    def _postfilter_trace(self, y, bm):
        """self.omlsa_multi as SubbandGSC.process(postfilter=True) leaves it (SubbandGSC.py:127-129,236-249).  Per block n of the call the
        reference analyses the block's output (transform_fbf, a streaming analysis) and the WHOLE bm_output array of the call as filled so
        far (transform_bm: blocks behind n are still zero, and the transform's carried overlap becomes the array's last block every time),
        takes frame 0 of that, and hands the two powers to NsOmlsaMulti.estimation — which reads the first M - 1 of the M blocking outputs.
        Reproduced with the native Transform and NsOmlsaMulti operators, block by block like the reference.  y [B, L], bm [B, M, L] float32."""
        from .ops import NsOmlsaMulti, Transform
        if not hasattr(self, "omlsa_multi"):
            dev = self._device
            self.omlsa_multi = NsOmlsaMulti(nfft=self.nfft, M=self.M, cal_weights=True, batch=self.batch, device=dev)      # :127
            self.transform_fbf = Transform(n_fft=self.nfft, hop_length=self.frameLen, channel=1, batch=self.batch, device=dev)   # :128
            self.transform_bm = Transform(n_fft=self.nfft, hop_length=self.frameLen, channel=self.M, batch=self.batch, device=dev)   # :129
        FL, B = self.frameLen, self.batch
        T = y.shape[1] // FL
        bm_t = np.ascontiguousarray(np.swapaxes(bm, 1, 2))                 # [B, L, M]
        # Frame 0 of the reference's whole-array analysis depends on the carried overlap and the array's FIRST block only, and the overlap it
        # leaves behind is the array's LAST block as filled so far (zeros until the final iteration): two one-block analyses per iteration
        # instead of one of the whole array — O(T) work per call where the literal restatement was O(T^2), same numbers bit for bit
        first = np.ascontiguousarray(bm_t[:, :FL])
        last_filled = np.ascontiguousarray(bm_t[:, (T - 1) * FL:])
        zeros = np.zeros_like(first)
        for n in range(T):
            Y = self.transform_fbf._eng.stft(np.ascontiguousarray(y[:, n * FL:(n + 1) * FL, None]), L.LAYOUT_SAMPLES_CHANNELS)   # [B, 1, K, 1]
            U = self.transform_bm._eng.stft(first, L.LAYOUT_SAMPLES_CHANNELS)                                                    # [B, 1, K, M]
            if T > 1:                                                       # the carried overlap := the array's last block (T = 1: it already is)
                self.transform_bm._eng.stft(last_filled if n == T - 1 else zeros, L.LAYOUT_SAMPLES_CHANNELS)
            yp = (Y[:, :1, :, 0].real.astype(np.float64) ** 2 + Y[:, :1, :, 0].imag.astype(np.float64) ** 2)
            up = (U[:, :1, :, : self.M - 1].real.astype(np.float64) ** 2 + U[:, :1, :, : self.M - 1].imag.astype(np.float64) ** 2)
            self.omlsa_multi.estimation_frames(yp, up)
