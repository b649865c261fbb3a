"""The two overlap-save GSCs of the reference, composed from the GPU operators.

  TDGSC   beamformer/TDGSC.py:24-175    time-aligned mean beamformer + pairwise-difference blocking matrix +
                                        MCRA-controlled multichannel FastFreqLms canceller (+ OMLSA post-filter)
  FDGSC   beamformer/FDGSC.py:38-317    time alignment + M coefficient-clamped adaptive blocking filters (mode 3) +
                                        norm-limited multichannel canceller (+ OMLSA post-filter)

Both structures are feed-forward from stage to stage (only state crosses blocks), so process() calls every operator ONCE with all
the blocks of the call: ds_dcnotch, ds_firbank_bm (FIR bank + channel mean + pairwise differences), ds_stft, ds_mcra_estimate_p,
ds_fdaf_update (all FFTs of a block inside one workgroup, the blocks walked in-kernel; FDGSC's M blocking filters are ONE batched
launch), ds_omlsa_postfilter (powers, gain and its application in-kernel), ds_istft.  This module sequences those calls, keeps the block delays (pure buffering) and derives the
adaptation-control scalars of FDGSC from the speech-presence matrix (mean p, the FDGSC.py:248-255 threshold) where the reference does."""
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

    def _spp(self, frames):
        """MCRA over the T frames of the call: complex [B, T, K] -> p [B, T, K] (mcra.p after each frame)."""
        return self.spp._eng.mcra_estimate_p(frames)[1].astype(np.float64)

    @staticmethod
    def _stft_refs(tf, bm):
        """STFT of the M-1 noise references [B, n, M-1] -> complex [B, T, K, M-1]."""
        return tf.stft(np.ascontiguousarray(bm, dtype=np.float32), L.LAYOUT_SAMPLES_CHANNELS)

    def _postfilter(self, out_td, U):
        """OMLSA gain on the canceller output (TDGSC.py:158-170 / FDGSC.py:286-298): out_td [B, n], U complex [B, T, K, M-1]
        (or [B, 1, K, M-1], the same references for every frame) -> post-filtered [B, n]."""
        Y = self.transform_fbf.stft(out_td[:, :, None], L.LAYOUT_SAMPLES_CHANNELS)[:, :, :, 0]            # [B, T, K]
        if U.shape[1] != Y.shape[1]:
            U = np.broadcast_to(U, (U.shape[0], Y.shape[1]) + U.shape[2:])
        _, Y = self.omlsa_multi._eng.omlsa_postfilter(Y, U)                      # powers, OMLSA gain and Y * sqrt(G) in the kernel
        return self.transform_fbf.istft(np.ascontiguousarray(Y[:, :, :, None]))[:, :, 0]


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
        self.transform_bm = BatchEngine(L.ALGO_TRANSFORM, M - 1, nb, frameLen, batch=B, device=device)      # :50

    def process(self, x, postfilter=False):
        """x [samples, chs] (or [B, samples, chs]) -> (output [samples], p [half_bin, blocks], output_bm [samples, chs-1])."""
        x, single = self._prep(x)
        x = np.swapaxes(self._notch.dcnotch(np.swapaxes(x, 1, 2)), 1, 2)                                   # :129-130
        if x.shape[1] == 0:
            z = np.zeros((self.batch, 0))
            out = (z, np.zeros((self.batch, self.frameLen + 1, 0)), np.zeros((self.batch, 0, self.M - 1)))
            return tuple(a[0] for a in out) if single else out
        xa, fixed, bm = self.time_alignment._eng.firbank(np.ascontiguousarray(x), want_bm=True)            # :143,149 all blocks
        D = self.transform.stft(fixed[:, :, None], L.LAYOUT_SAMPLES_CHANNELS)[:, :, :, 0]                   # :145  [B, T, K]
        p = self._spp(D)                                                                                    # :146-147
        out, w = self.aic_filter._eng.fdaf_update(bm, fixed, p=p, fir_truncate=30, p_complement=True)       # :152-156 -> :105 (p = 1 - p)
        self.aic_filter._w = w.astype(np.float64)
        if postfilter:                                                                                      # :158-170
            out = self._postfilter(out, self._stft_refs(self.transform_bm, bm))
        res = (out.astype(np.float64), np.swapaxes(p, 1, 2), bm.astype(np.float64))
        return tuple(a[0] for a in res) if single else res


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
        self.transform_bm = BatchEngine(L.ALGO_TRANSFORM, M - 1, nb, frameLen, batch=B, device=device)      # :110
        self._fix_prev = np.zeros((B, frameLen), dtype=np.float32)             # delay_fbf: one block (:93)
        self._al_tail = np.zeros((B, frameLen // 2, M), dtype=np.float32)      # delay_aligned: half a block (:96)
        self._bm_last = np.zeros((B, frameLen, M - 1), dtype=np.float32)       # last hop of the array transform_bm saw last
        self._tf_u = BatchEngine(L.ALGO_TRANSFORM, M - 1, nb, frameLen, batch=B, device=device)

    def process(self, x, postfilter=False, dc_notch=True):
        """x [samples, chs] (or [B, samples, chs]) -> (output, p, fix_output, fix_output_delayed, bm_output,
        aligned_output, aligned_output_delayed)."""
        x, single = self._prep(x)
        B, M, FL = self.batch, self.M, self.frameLen
        K, H = FL + 1, FL // 2
        if dc_notch:
            x = np.swapaxes(self._notch.dcnotch(np.swapaxes(x, 1, 2)), 1, 2)                               # :213-215
        x = np.ascontiguousarray(x)
        ns = x.shape[1]
        nblk = ns // FL
        if nblk == 0:
            z1, zM = np.zeros((B, 0)), np.zeros((B, 0, M))
            out = (z1, np.zeros((B, K, 0)), z1, z1, zM, zM, zM)
            return tuple(a[0] for a in out) if single else out
        xa, fixed = self.time_alignment._eng.firbank(x)                                                     # :235,238  all blocks
        D = self.transform_x.stft(x, L.LAYOUT_SAMPLES_CHANNELS)[:, :, :, 0]                                  # :241 (channel 0, mcra.py:32-33)
        p = self._spp(D)                                                                                    # :243-244  [B, T, K]
        hot = np.mean(p[:, :, 32:128], axis=2) > 0.8                                                        # :248-255, per block
        lo = p[:, :, :32]
        lo[hot[:, :, None] & (lo < 0.8)] = 0.8
        xad = np.concatenate((self._al_tail, xa[:, : ns - H]), axis=1)                                      # :258 delay_aligned
        self._al_tail = xa[:, ns - H:].copy()
        # :259-264 -> :185-195: M filters, input = fixed beamformer output, desired = delayed aligned channel m, p = 1
        xin = np.repeat(fixed[:, None, :], M, axis=1).reshape(B * M, ns, 1)
        din = np.ascontiguousarray(np.swapaxes(xad, 1, 2)).reshape(B * M, ns)
        e_bm, w_bm = self.bm._eng.fdaf_update(xin, din)
        self.bm._w = w_bm.astype(np.float64)
        bm_output = np.ascontiguousarray(np.swapaxes(e_bm.reshape(B, M, ns), 1, 2))                         # [B, n, M]
        fix_d = np.concatenate((self._fix_prev, fixed[:, : ns - FL]), axis=1)                               # :270 delay_fbf
        self._fix_prev = fixed[:, ns - FL:].copy()
        pa = 1.0 - np.mean(p, axis=2)                                                                       # :282  [B, T]
        out, w = self.aic_filter._eng.fdaf_update(bm_output, fix_d, p=pa)                                   # :278-284
        self.aic_filter._w = w.astype(np.float64)
        if postfilter:                                                                                      # :286-298
            # transform_fbf is shared by the delayed fixed output (:273) and the canceller output (:287): its STFT state alternates
            # between the two signals block by block, i.e. frame t of either analysis starts from the OTHER signal's previous block
            out = self._postfilter_fdgsc(out, fix_d, bm_output)
        else:
            self.transform_fbf.stft(fix_d[:, :, None], L.LAYOUT_SAMPLES_CHANNELS)                           # :273 keeps advancing the shared state
        res = (out.astype(np.float64), np.swapaxes(p, 1, 2), fixed.astype(np.float64), fix_d.astype(np.float64),
               bm_output.astype(np.float64), xa.astype(np.float64), xad.astype(np.float64))
        return tuple(a[0] for a in res) if single else res

    def _postfilter_fdgsc(self, out, fix_d, bm_output):
        """FDGSC.py:273,286-298 block by block: the shared transform_fbf and the whole-array re-analysis of bm_output make this
        branch inherently sequential in the reference; each block is three kernel calls here."""
        B, M, FL = self.batch, self.M, self.frameLen
        nblk = out.shape[1] // FL
        res = np.empty_like(out)
        U0 = None
        for n in range(nblk):
            sl = slice(n * FL, (n + 1) * FL)
            self.transform_fbf.stft(np.ascontiguousarray(fix_d[:, sl, None]), L.LAYOUT_SAMPLES_CHANNELS)   # :273 (advances the shared state)
            # the reference re-analyses the WHOLE bm_output array every block and keeps frame 0 (:288,291): that frame is
            # [last hop of the array at the previous call | block 0 of this array] — constant within one process() call
            if n <= 1:
                prev = self._bm_last if n == 0 else np.zeros_like(self._bm_last)
                self._tf_u.reset()
                self._stft_refs(self._tf_u, prev)
                U0 = self._stft_refs(self._tf_u, bm_output[:, :FL, :-1])
            res[:, sl] = self._postfilter(np.ascontiguousarray(out[:, sl]), U0)
        self._bm_last = bm_output[:, -FL:, :-1].astype(np.float32)
        return res
