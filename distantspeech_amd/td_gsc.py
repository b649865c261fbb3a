"""The two overlap-save GSCs of the reference, each behind ONE native chain handle.

  TDGSC   beamformer/TDGSC.py:24-175    time-aligned mean beamformer + pairwise-difference blocking matrix +
                                        MCRA-controlled multichannel FastFreqLms canceller (+ OMLSA post-filter)       DS_ALGO_TDGSC
  FDGSC   beamformer/FDGSC.py:38-317    time alignment + M coefficient-clamped adaptive blocking filters (mode 3) +
                                        norm-limited multichannel canceller (+ OMLSA post-filter)                      DS_ALGO_FDGSC

process() is one call into libdsenh.so (csrc/ds_api_gsc_chains.hip): DC notch, FIR bank + channel mean (+ pairwise differences),
analysis, MCRA, the overlap-save filters, the block delays, FDGSC's adaptation control (mean p, the FDGSC.py:248-255 threshold) and the
OMLSA post-filter all run on the device, every stage on the previous stage's buffer; this module only reshapes the arguments and the
returned tuple to the reference's layout."""
import numpy as np

from . import _lib as L
from .engine import BatchEngine
from .mic_array import MicArray, compute_tau
from .subband_gsc import fractional_delay_filter_bank


class _Weights(object):
    """stand-in for the reference's filter attributes that callers read back (`aic_filter.w`, `bm[m].w`)."""

    def __init__(self, shape):
        self.w = np.zeros(shape)


class _BlockGSC(object):
    def _setup(self, algo, mic_array, frameLen, angle, batch, device):
        self.mic_array, self.M, self.frameLen, self.batch = mic_array, mic_array.M, frameLen, int(batch)
        self.angle = np.array(angle) / 180 * np.pi if isinstance(angle, list) else angle
        tau = compute_tau(mic_array, self.angle)
        self.delay_filter = fractional_delay_filter_bank(np.array(-(tau - np.max(tau)))[:, 0] * mic_array.fs)   # fixedbeamformer.py:67-70
        self._eng = BatchEngine(algo, self.M, 2 * frameLen, frameLen, batch=batch, device=device)
        self._eng.chain_set_aux(L.CHAIN_AUX_FIR, self.delay_filter)

    def _prep(self, x):
        x = np.asarray(x)
        single = x.ndim == 2
        if single:
            if self.batch != 1:
                raise ValueError("object built with batch=%d; pass [B, n_samples, n_chs]" % self.batch)
            x = x[None]
        if x.shape[2] != self.M or x.shape[1] % self.frameLen != 0:
            raise ValueError("x must be [k * %d samples, n_chs=%d]" % (self.frameLen, self.M))
        return np.ascontiguousarray(np.swapaxes(x, 1, 2), dtype=np.float32), single       # the chains take [B, M, n]


class TDGSC(_BlockGSC):
    """Time-domain GSC — beamformer/TDGSC.py:24-175."""

    def __init__(self, mic_array: MicArray, frameLen=256, angle=[197, 0], batch=1, device=-1):
        self._setup(L.ALGO_TDGSC, mic_array, frameLen, angle, batch, device)
        self.aic_filter = _Weights((frameLen, self.M - 1))

    def process(self, x, postfilter=False):
        """x [samples, chs] (or [B, samples, chs]) -> (output [samples], p [half_bin, blocks], output_bm [samples, chs-1])."""
        x, single = self._prep(x)
        if x.shape[2] == 0:
            z = np.zeros((self.batch, 0))
            out = (z, np.zeros((self.batch, self.frameLen + 1, 0)), np.zeros((self.batch, 0, self.M - 1)))
            return tuple(a[0] for a in out) if single else out
        out, p, bm, w = self._eng.tdgsc_process(x, postfilter=postfilter)
        self.aic_filter.w = (w[0] if single else w).astype(np.float64)
        res = (out.astype(np.float64), np.swapaxes(p, 1, 2).astype(np.float64), bm.astype(np.float64))
        return tuple(a[0] for a in res) if single else res


class FDGSC(_BlockGSC):
    """Overlap-save frequency-domain GSC with adaptive blocking matrix (mode 3) — beamformer/FDGSC.py:38-317."""

    def __init__(self, mic_array: MicArray, frameLen=256, angle=[197, 0], batch=1, device=-1):
        self._setup(L.ALGO_FDGSC, mic_array, frameLen, angle, batch, device)
        self.nfft = 2 * frameLen
        self.aic_filter = _Weights((frameLen, self.M))
        self.bm = _Weights((self.batch * self.M, frameLen, 1))

    def process(self, x, postfilter=False, dc_notch=True):
        """x [samples, chs] (or [B, samples, chs]) -> (output, p, fix_output, fix_output_delayed, bm_output,
        aligned_output, aligned_output_delayed)."""
        x, single = self._prep(x)
        B, M, K = self.batch, self.M, self.frameLen + 1
        if x.shape[2] == 0:
            z1, zM = np.zeros((B, 0)), np.zeros((B, 0, M))
            out = (z1, np.zeros((B, K, 0)), z1, z1, zM, zM, zM)
            return tuple(a[0] for a in out) if single else out
        r = self._eng.fdgsc_process(x, postfilter=postfilter, dc_notch=dc_notch)
        self.aic_filter.w = (r["w_aic"][0] if single else r["w_aic"]).astype(np.float64)
        self.bm.w = r["w_bm"][:, :, None].astype(np.float64)
        t = lambda a: np.swapaxes(a, 1, 2).astype(np.float64)                                   # [B, M, n] -> [B, n, M]
        res = (r["out"].astype(np.float64), t(r["p"]), r["fix"].astype(np.float64), r["fix_d"].astype(np.float64), t(r["bm"]), t(r["al"]),
               t(r["al_d"]))
        return tuple(a[0] for a in res) if single else res
