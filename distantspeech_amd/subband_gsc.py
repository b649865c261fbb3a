"""SubbandGSC (BASELINE config-5 structure) and its front-end conditioning, composed from the GPU operators.

  fractional_delay_filter_bank   transform/multirate.py:4-51        (host set-up, once per look direction)
  FilterDcNotch16                adaptivefilter/feature.py:32-49    (GPU: ds_dcnotch)
  TimeAlignment                  beamformer/fixedbeamformer.py:51-93 (GPU: ds_firbank)
  DelaySamples                   beamformer/utils.py:241-274        (a buffer; no arithmetic)
  SubbandGSC                     beamformer/SubbandGSC.py:67-262

Every arithmetic stage of SubbandGSC.process runs in a libdsenh kernel (FIR bank + channel mean, STFT, McSpp,
the M adaptive blocking filters as ONE batched subband-LMS/RLS launch, ISTFT, re-analysis, multichannel canceller,
ISTFT); this module only sequences the calls and moves buffers.  A single fused kernel for this composition is
planned; the operator-level composition is the parity-first version."""
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
        return out.astype(np.float64), self.notch_mem


class DelaySamples(object):
    """beamformer/utils.py:241-274 — pure buffering."""

    def __init__(self, data_len, delay, channel=1, dtype=np.float64):
        self.data_len, self.n_delay = data_len, delay
        self.buffer = np.zeros(((data_len + delay), channel), dtype=dtype)

    def delay(self, x):
        if len(x.shape) == 1:
            x = x[:, np.newaxis]
        data_len = x.shape[0]
        if self.n_delay == 0:
            return x
        self.buffer[-data_len:, :] = x
        output = self.buffer[:data_len, :].copy()
        self.buffer[: self.n_delay, :] = self.buffer[-self.n_delay:, :]
        return output


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

    `bm_filter="rls"` swaps the blocking filters for SubbandRLS(filter_len=2) — the BASELINE config-5 composition
    (defined by us, the reference never composes it)."""

    def __init__(self, mic_array: MicArray, frameLen=256, angle=[197, 0], batch=1, device=-1, bm_filter="lms"):
        self.M, self.frameLen, self.batch = mic_array.M, frameLen, int(batch)
        self.MicArray = mic_array
        M, B, nb = self.M, self.batch, 2 * frameLen
        self.nfft, self.hop, self.half_bin = nb, frameLen, frameLen + 1
        self.angle = np.array(angle) / 180 * np.pi if isinstance(angle, list) else angle
        self.time_alignment = TimeAlignment(mic_array, angle=self.angle, batch=B, device=device)           # :85
        self._notch = BatchEngine(L.ALGO_FRONTEND, M, nb, batch=B, device=device, filt_alpha=0.98)          # :122-124
        self.transform = BatchEngine(L.ALGO_TRANSFORM, M, nb, frameLen, batch=B, device=device)             # :117
        self.spp = McSpp(nfft=nb, channels=M, batch=B, device=device)                                        # :115
        self._tf_fixed = BatchEngine(L.ALGO_TRANSFORM, 1, nb, frameLen, batch=B, device=device)             # bm[m].transform_x (identical for all m)
        self.bm_filter = bm_filter
        if bm_filter == "rls":
            self._bm = BatchEngine(L.ALGO_SUBRLS, 1, nb, batch=B * M, device=device, filter_len=2)
        else:
            self._bm = BatchEngine(L.ALGO_SUBLMS, 1, nb, batch=B * M, device=device, filter_len=2, filt_mu=1e-1)   # :99-101
        self._tf_bm = BatchEngine(L.ALGO_TRANSFORM, M, nb, frameLen, batch=B, device=device)                # bm[m].transform_d synthesis
        self._tf_aic_x = BatchEngine(L.ALGO_TRANSFORM, M, nb, frameLen, batch=B, device=device)             # aic_filter.transform_x
        self._aic = BatchEngine(L.ALGO_SUBLMS, M, nb, batch=B, device=device, filter_len=2, filt_mu=0.01, filt_alpha=0.8)   # :103-109
        self._tf_aic_d = BatchEngine(L.ALGO_TRANSFORM, 1, nb, frameLen, batch=B, device=device)             # aic_filter.transform_d
        self._F_prev = np.zeros((B, self.half_bin), dtype=np.complex64)     # STFT of the fixed output delayed by one block (:111,226)
        self._fix_prev = np.zeros((B, frameLen), dtype=np.float32)

    def fixed_beamformer(self, x):
        return np.mean(x, axis=1, keepdims=True)

    def process(self, x, postfilter=False):
        """x [n_chs, n_samples] (or [B, n_chs, n_samples]) ->
        (output [L], fix_output [L], bm_output [L, M], p [half_bin, blocks], aligned_output [L, M])."""
        if postfilter:
            # the reference's post-filter branch re-analyses the WHOLE bm_output array every block (SubbandGSC.py:238)
            # and never feeds its gain back into the returned signal (:248 commented out)
            raise NotImplementedError("postfilter=True does not change the reference's output and is not built")
        x = np.asarray(x)
        single = x.ndim == 2
        if single:
            if self.batch != 1:
                raise ValueError("object built with batch=%d; pass [B, n_chs, n_samples]" % self.batch)
            x = x[None]
        B, M, FL, K = self.batch, self.M, self.frameLen, self.half_bin
        if x.shape[1] != M or x.shape[2] % FL != 0:
            raise ValueError("x must be [n_chs=%d, k * %d samples]" % (M, FL))
        x = self._notch.dcnotch(x)                                              # :177-178
        nblk = x.shape[2] // FL
        output = np.zeros((B, nblk * FL)); fix_output = np.zeros((B, nblk * FL))
        bm_output = np.zeros((B, nblk * FL, M)); aligned = np.zeros((B, nblk * FL, M))
        p = np.zeros((B, K, nblk))
        for n in range(nblk):
            sl = slice(n * FL, (n + 1) * FL)
            xa, fixed = self.time_alignment.process_with_mean(np.ascontiguousarray(np.swapaxes(x[:, :, sl], 1, 2)))   # :201,206
            aligned[:, sl] = xa
            D = self.transform.stft(xa, L.LAYOUT_SAMPLES_CHANNELS)[:, 0]        # [B, K, M]   :204
            pn = self.spp._eng.mcspp_estimate(D[:, None], want_yout=False)["p"][:, 0]   # :208
            p[:, :, n] = pn
            F = self._tf_fixed.stft(fixed[:, :, None], L.LAYOUT_SAMPLES_CHANNELS)[:, 0, :, 0]   # [B, K]
            # M adaptive blocking filters as one batched launch: utterance-major, filter-minor   :217-223
            xin = np.repeat(F[:, None, :], M, axis=1).reshape(B * M, 1, K)
            din = np.ascontiguousarray(np.swapaxes(D, 1, 2)).reshape(B * M, 1, K)
            if self.bm_filter == "rls":
                err = self._bm.subrls_update(xin, din)
            else:
                pin = np.repeat(pn[:, None, :], M, axis=1).reshape(B * M, 1, K)
                err = self._bm.sublms_update(xin[..., None], din, pin)
            E = np.ascontiguousarray(np.swapaxes(err.reshape(B, M, K), 1, 2))   # [B, K, M]
            bm_td = self._tf_bm.istft(E[:, None])                               # [B, FL, M]
            bm_output[:, sl] = bm_td
            Xa = self._tf_aic_x.stft(bm_td, L.LAYOUT_SAMPLES_CHANNELS)[:, 0]    # [B, K, M]   :230-234
            Dd = self._F_prev                                                   # analysis of the block-delayed fixed output
            e2 = self._aic.sublms_update(Xa[:, None], Dd[:, None], (1.0 - pn)[:, None])
            out_td = self._tf_aic_d.istft(e2[:, :, :, None])[:, :, 0]
            output[:, sl] = out_td
            fix_output[:, sl] = self._fix_prev
            self._F_prev, self._fix_prev = F, fixed
        sq = (lambda a: a[0]) if single else (lambda a: a)
        return sq(output), sq(fix_output), sq(bm_output), sq(p), sq(aligned)
