"""BASELINE config 4: WPE dereverberation -> MCRA-gated adaptive MVDR -> multichannel-SPP gain, all on one STFT grid.

The reference has the three pieces (Wpe.update dereverberation/awpe.py:129-192, adaptivebeamfomer.process
beamformer/adaptivebeamformer.py:44-128, the McMcra gain post-filter of GSC.py:225,286) but no class that chains them;
the chain is defined the way GSC.process chains its stages (one STFT, per-frame stages on the spectrum, one ISTFT) and
lives behind ONE native handle (DS_ALGO_WPE_MVDR): ds_process() runs STFT -> frame delay line -> ds_wpe_update (all channels)
-> ds_mcmcra_estimate -> ds_adaptive_frames (gain applied in the kernel) -> ISTFT as six launches on the handle's stream,
device-resident between the stages, all T hops of a call per launch.  This class only mirrors adaptivebeamfomer's interface."""
import numpy as np

from . import _lib as L
from .engine import BatchEngine
from .mic_array import MicArray


class WpeMvdrPostfilter(object):
    def __init__(self, mic_array: MicArray, frameLen=1024, hop=None, nfft=None, taps=2, delay=4, forgetting_factor=0.998,
                 mcra_L=15, batch=1, device=-1):
        self.MicArray, self.M, self.batch = mic_array, mic_array.M, int(batch)
        self.nfft = int(nfft) if nfft else int(frameLen)
        self.hop = int(hop) if hop else self.nfft // 2
        self.half_bin = self.nfft // 2 + 1
        self._eng = BatchEngine(L.ALGO_WPE_MVDR, self.M, self.nfft, self.hop, batch=batch, device=device, filter_len=taps,
                                rls_lambda=forgetting_factor, mcra_L=mcra_L)
        self._eng.set_mcra_L(mcra_L)
        self._eng.set_wpe_delay(delay)
        self._angle = None

    def _steer(self, angle):
        """a[k, m] = exp(-j w_k tao_m), circular-array tao — adaptivebeamformer.py:52,84."""
        mic = self.MicArray
        angle = np.asarray(angle, dtype=float)
        tao = -1 * mic.r * np.cos(angle[1]) * np.cos(angle[0] - mic.gamma) / mic.c
        omega = 2 * np.pi * np.arange(self.half_bin) * mic.fs / self.nfft
        return np.exp(-1j * omega[:, None] * tao[None, :])

    def process(self, x, angle, method=2):
        """x [n_chs, T*hop] (or [B, n_chs, T*hop]) -> {'data': y [T*hop]} like adaptivebeamfomer.process."""
        x = np.asarray(x)
        single = x.ndim == 2
        if single:
            if self.batch != 1:
                raise ValueError("object built with batch=%d; pass [B, n_chs, n_samples]" % self.batch)
            x = x[None]
        if x.shape[1] != self.M or x.shape[2] % self.hop != 0:
            raise ValueError("x must be [n_chs=%d, k * %d samples]" % (self.M, self.hop))
        key = tuple(np.asarray(angle, dtype=float).ravel())
        if key != self._angle:
            self._eng.set_steering(self._steer(angle))
            self._angle = key
        self._eng.set_method(method)
        y = self._eng.process(x, L.LAYOUT_CHANNELS_SAMPLES).astype(np.float64)
        return {"data": y[0] if single else y}
