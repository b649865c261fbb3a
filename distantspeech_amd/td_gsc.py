"""The two overlap-save GSCs of the reference, composed from the GPU operators.

  TDGSC   beamformer/TDGSC.py:24-175    time-aligned mean beamformer + pairwise-difference blocking matrix +
                                        MCRA-controlled multichannel FastFreqLms canceller (+ OMLSA post-filter)
  FDGSC   beamformer/FDGSC.py:38-317    time alignment + M coefficient-clamped adaptive blocking filters (mode 3) +
                                        norm-limited multichannel canceller (+ OMLSA post-filter)

Every signal-path stage runs in a libdsenh kernel: ds_dcnotch, ds_firbank_bm (FIR bank + channel mean + pairwise
differences), ds_stft / ds_istft, ds_mcra_estimate, ds_fdaf_update (all FFTs of a block inside one workgroup; the M
blocking filters of FDGSC as ONE batched launch), ds_omlsa_estimate.  This module sequences the calls, keeps the
block delays (pure buffering) and derives the adaptation-control values from the speech-presence vector
(1 - p, mean p, the FDGSC.py:248-255 threshold) exactly where the reference does."""
import numpy as np

from . import _lib as L
from .engine import BatchEngine
from .mic_array import MicArray
from .ops import AdaptiveBlockingMatrixFilter, AdaptiveInterferenceCancellation, FastFreqLms, NoiseEstimationMCRA, NsOmlsaMulti
from .subband_gsc import TimeAlignment


class _BlockGSC(object):
    def _prep(self, x):
        x = np.asarray(x)
        single = x.ndim == 2
        if single:
            if self.batch != 1:
                raise ValueError("object built with batch=%d; pass [B, n_samples, n_chs]" % self.batch)
            x = x[None]
        if x.shape[2] != self.M or x.shape[1] % self.frameLen != 0:
            raise ValueError("x must be [k * %d samples, n_chs=%d]" % (self.frameLen, self.M))
        return x, single

    def _spp_block(self, frame):
        """one MCRA step on a complex frame [B, K] -> p [B, K]."""
        self.spp._eng.mcra_estimate(frame[:, None, :])
        return self.spp._eng.op_state()[:, 3, :].astype(np.float64)

    @staticmethod
    def _stft_refs(tf, bm):
        """STFT of the M-1 noise references [B, n, M-1] on an M-channel transform handle (the kernels are built for
        even channel counts): the last channel is fed zeros and dropped."""
        pad = np.zeros(bm.shape[:2] + (1,), dtype=np.float32)
        return tf.stft(np.concatenate((bm.astype(np.float32), pad), axis=2), L.LAYOUT_SAMPLES_CHANNELS)[:, :, :, :-1]

    def _postfilter(self, out_td, U):
        """OMLSA gain on the canceller output (TDGSC.py:158-170 / FDGSC.py:286-298): out_td [B, FL], U complex [B, K, M-1]."""
        Y = self.transform_fbf.stft(out_td[:, :, None], L.LAYOUT_SAMPLES_CHANNELS)[:, 0, :, 0]              # [B, K]
        y_pow = (Y.real.astype(np.float64) ** 2 + Y.imag.astype(np.float64) ** 2)
        u_pow = (U.real.astype(np.float64) ** 2 + U.imag.astype(np.float64) ** 2)
        _, G, _ = self.omlsa_multi._eng.omlsa_estimate(y_pow[:, None], u_pow[:, None])
        Y = Y * np.sqrt(G[:, 0].astype(np.float64))
        return self.transform_fbf.istft(np.ascontiguousarray(Y[:, None, :, None]))[:, :, 0]


class TDGSC(_BlockGSC):
    """Time-domain GSC — beamformer/TDGSC.py:24-175."""

    def __init__(self, mic_array: MicArray, frameLen=256, angle=[197, 0], batch=1, device=-1):
        self.mic_array, self.M, self.frameLen, self.batch = mic_array, mic_array.M, frameLen, int(batch)
        M, B, nb = self.M, self.batch, 2 * frameLen
        self.angle = np.array(angle) / 180 * np.pi if isinstance(angle, list) else angle
        self.time_alignment = TimeAlignment(mic_array, angle=self.angle, batch=B, device=device)            # :36
        self.aic_filter = FastFreqLms(filter_len=frameLen, n_channels=M - 1, non_causal=True, batch=B, device=device)   # :37
        self._notch = BatchEngine(L.ALGO_FRONTEND, M, nb, batch=B, device=device, filt_alpha=0.98)          # :38-40
        self.mcra = NoiseEstimationMCRA(nfft=nb, batch=B, device=device)                                    # :42-43
        self.mcra.L = 65
        self.spp = self.mcra                                                                                # :46
        self.transform = BatchEngine(L.ALGO_TRANSFORM, 1, nb, frameLen, batch=B, device=device)             # :44
        self.omlsa_multi = NsOmlsaMulti(nfft=nb, cal_weights=True, M=M, batch=B, device=device)             # :48
        self.transform_fbf = BatchEngine(L.ALGO_TRANSFORM, 1, nb, frameLen, batch=B, device=device)         # :49
        self.transform_bm = BatchEngine(L.ALGO_TRANSFORM, M, nb, frameLen, batch=B, device=device)          # :50 (M-1 used)

    def process(self, x, postfilter=False):
        """x [samples, chs] (or [B, samples, chs]) -> (output [samples], p [half_bin, blocks], output_bm [samples, chs-1])."""
        x, single = self._prep(x)
        B, M, FL = self.batch, self.M, self.frameLen
        K = FL + 1
        x = np.swapaxes(self._notch.dcnotch(np.swapaxes(x, 1, 2)), 1, 2)                                   # :129-130
        nblk = x.shape[1] // FL
        output = np.zeros((B, nblk * FL)); output_bm = np.zeros((B, nblk * FL, M - 1)); p = np.zeros((B, K, nblk))
        for n in range(nblk):
            sl = slice(n * FL, (n + 1) * FL)
            xa, fixed, bm = self.time_alignment._eng.firbank(np.ascontiguousarray(x[:, sl]), want_bm=True)  # :143,149
            D = self.transform.stft(fixed[:, :, None], L.LAYOUT_SAMPLES_CHANNELS)[:, 0, :, 0]               # :145
            pn = self._spp_block(D)                                                                         # :146-147
            p[:, :, n] = pn
            out_n, w = self.aic_filter._eng.fdaf_update(bm, fixed, p=(1.0 - pn)[:, None, :], fir_truncate=30)   # :152-156 -> :105
            self.aic_filter._w = w.astype(np.float64)
            if postfilter:                                                                                  # :158-170
                U = self._stft_refs(self.transform_bm, bm)[:, 0]
                out_n = self._postfilter(out_n, U)
            output_bm[:, sl] = bm
            output[:, sl] = out_n
        sq = (lambda a: a[0]) if single else (lambda a: a)
        return sq(output), sq(p), sq(output_bm)


class FDGSC(_BlockGSC):
    """Overlap-save frequency-domain GSC with adaptive blocking matrix (mode 3) — beamformer/FDGSC.py:38-317."""

    def __init__(self, mic_array: MicArray, frameLen=256, angle=[197, 0], batch=1, device=-1):
        self.mic_array, self.M, self.frameLen, self.batch = mic_array, mic_array.M, frameLen, int(batch)
        M, B, nb = self.M, self.batch, 2 * frameLen
        self.nfft = nb
        self.angle = np.array(angle) / 180 * np.pi if isinstance(angle, list) else angle
        self.time_alignment = TimeAlignment(mic_array, angle=self.angle, batch=B, device=device)            # :56
        # the M blocking filters (:71-81) as one batch of B * M single-channel instances: utterance-major, filter-minor
        self.bm = AdaptiveBlockingMatrixFilter(filter_len=frameLen, mu=0.1, alpha=0.9, non_causal=False, constrain=True,
                                               batch=B * M, device=device)
        self.aic_filter = AdaptiveInterferenceCancellation(filter_len=frameLen, n_channels=M, mu=0.1, alpha=0.9, non_causal=False,
                                                           constrain=True, weight_norm=True, batch=B, device=device)   # :83-91
        self._notch = BatchEngine(L.ALGO_FRONTEND, M, nb, batch=B, device=device, filt_alpha=0.98)          # :114-116
        self.spp = NoiseEstimationMCRA(nfft=nb, batch=B, device=device)                                     # :99-100
        self.spp.L = 60
        self.transform_x = BatchEngine(L.ALGO_TRANSFORM, M, nb, frameLen, batch=B, device=device)           # :106
        self.omlsa_multi = NsOmlsaMulti(nfft=nb, cal_weights=True, M=M, batch=B, device=device)             # :108
        self.transform_fbf = BatchEngine(L.ALGO_TRANSFORM, 1, nb, frameLen, batch=B, device=device)         # :109
        self.transform_bm = BatchEngine(L.ALGO_TRANSFORM, M, nb, frameLen, batch=B, device=device)          # :110 (M-1 used)
        self._fix_prev = np.zeros((B, frameLen), dtype=np.float32)             # delay_fbf: one block (:93)
        self._al_tail = np.zeros((B, frameLen // 2, M), dtype=np.float32)      # delay_aligned: half a block (:96)
        self._bm_last = np.zeros((B, frameLen, M - 1), dtype=np.float32)       # last hop of the array transform_bm saw last
        self._tf_u = BatchEngine(L.ALGO_TRANSFORM, M, nb, frameLen, batch=B, device=device)

    def process(self, x, postfilter=False, dc_notch=True):
        """x [samples, chs] (or [B, samples, chs]) -> (output, p, fix_output, fix_output_delayed, bm_output,
        aligned_output, aligned_output_delayed)."""
        x, single = self._prep(x)
        B, M, FL = self.batch, self.M, self.frameLen
        K, H = FL + 1, FL // 2
        if dc_notch:
            x = np.swapaxes(self._notch.dcnotch(np.swapaxes(x, 1, 2)), 1, 2)                               # :213-215
        nblk = x.shape[1] // FL
        ns = nblk * FL
        output = np.zeros((B, ns)); bm_output = np.zeros((B, ns, M)); p = np.zeros((B, K, nblk))
        aligned = np.zeros((B, ns, M)); aligned_d = np.zeros((B, ns, M)); fix = np.zeros((B, ns)); fix_d = np.zeros((B, ns))
        U0 = None
        for n in range(nblk):
            sl = slice(n * FL, (n + 1) * FL)
            xn = np.ascontiguousarray(x[:, sl])
            xa, fixed = self.time_alignment._eng.firbank(xn)                                                # :235,238
            D = self.transform_x.stft(xn, L.LAYOUT_SAMPLES_CHANNELS)[:, 0, :, 0]                            # :241 (channel 0, mcra.py:32-33)
            pn = self._spp_block(D)                                                                         # :243-244
            for b in range(B):                                                                              # :248-255
                if np.mean(pn[b, 32:128]) > 0.8:
                    lo = pn[b, :32]
                    lo[lo < 0.8] = 0.8
            p[:, :, n] = pn
            xad = np.concatenate((self._al_tail, xa[:, : FL - H]), axis=1)                                  # :258
            self._al_tail = xa[:, FL - H:].copy()
            # :259-264 -> :185-195: M filters, input = fixed beamformer output, desired = delayed aligned channel m, p = 1
            xin = np.repeat(fixed[:, None, :], M, axis=1).reshape(B * M, FL, 1)
            din = np.ascontiguousarray(np.swapaxes(xad, 1, 2)).reshape(B * M, FL)
            e_bm, w_bm = self.bm._eng.fdaf_update(xin, din)
            self.bm._w = w_bm.astype(np.float64)
            bm_n = np.ascontiguousarray(np.swapaxes(e_bm.reshape(B, M, FL), 1, 2))                          # [B, FL, M]
            bm_output[:, sl] = bm_n
            fixed_dn = self._fix_prev                                                                       # :270
            self._fix_prev = fixed
            self.transform_fbf.stft(fixed_dn[:, :, None], L.LAYOUT_SAMPLES_CHANNELS)                        # :273 (advances the shared state)
            pa = (1.0 - np.mean(pn, axis=1))[:, None]                                                       # :282
            out_n, w = self.aic_filter._eng.fdaf_update(bm_n, fixed_dn, p=pa)                               # :278-284
            self.aic_filter._w = w.astype(np.float64)
            if postfilter:                                                                                  # :286-298
                # the reference re-analyses the WHOLE bm_output array every block and keeps frame 0 (:288,291): that frame is
                # [last hop of the array at the previous call | block 0 of this array] — constant within one process() call
                if n <= 1:
                    prev = self._bm_last if n == 0 else np.zeros_like(self._bm_last)
                    self._tf_u.reset()
                    self._stft_refs(self._tf_u, prev)
                    U0 = self._stft_refs(self._tf_u, bm_output[:, :FL, :-1])[:, 0]
                out_n = self._postfilter(out_n, U0)
            fix[:, sl] = fixed; fix_d[:, sl] = fixed_dn
            aligned[:, sl] = xa; aligned_d[:, sl] = xad
            output[:, sl] = out_n
        if postfilter and nblk > 0:
            self._bm_last = bm_output[:, -FL:, :-1].astype(np.float32)
        sq = (lambda a: a[0]) if single else (lambda a: a)
        return sq(output), sq(p), sq(fix), sq(fix_d), sq(bm_output), sq(aligned), sq(aligned_d)
