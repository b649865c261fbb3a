"""distantspeech_amd — MI355X-native per-frame multichannel speech enhancement
(streaming STFT -> beamformer -> post-filter -> ISTFT overlap-add), a drop-in for that hot path of
wangwei2009/DistantSpeech.  The compute lives in libdsenh.so (hand-written HIP for gfx950,
C-ABI in include/dsenh.h); this package mirrors the reference's Python object interface on top."""
from . import _lib
from .engine import BatchEngine
from .mic_array import MicArray, compute_tau, gen_noise_msc
from .beamformer import beamformer, FixedBeamformer, adaptivebeamfomer, GSC, compute_mvdr_weight
from .ops import Transform, NoiseEstimationMCRA, McMcra, McSppBase, McSpp, OnlineMvdr, steering, compute_pmwf_weight, get_gev_vector, blind_analytic_normalization, phase_correction, NsOmlsaMulti, SubbandLMS, SubbandLmsMc, SubbandRLS, Wpe, BaseFilter, Rls, FastFreqLms, AdaptiveBlockingMatrixFilter, AdaptiveInterferenceCancellation

from .subband_gsc import SubbandGSC, TimeAlignment, FilterDcNotch16, DelaySamples, fractional_delay_filter_bank
from .td_gsc import TDGSC, FDGSC
from .dereverb_mvdr import WpeMvdrPostfilter

__all__ = ["SubbandGSC", "TimeAlignment", "FilterDcNotch16", "DelaySamples", "fractional_delay_filter_bank", "BatchEngine", "MicArray", "compute_tau", "gen_noise_msc", "beamformer", "FixedBeamformer",
           "adaptivebeamfomer", "GSC", "compute_mvdr_weight", "Transform", "NoiseEstimationMCRA", "McMcra", "McSppBase", "McSpp", "OnlineMvdr", "steering", "compute_pmwf_weight", "get_gev_vector", "blind_analytic_normalization", "phase_correction", "NsOmlsaMulti",
           "SubbandLMS", "SubbandLmsMc", "SubbandRLS", "Wpe", "BaseFilter", "Rls", "FastFreqLms", "AdaptiveBlockingMatrixFilter", "AdaptiveInterferenceCancellation",
           "TDGSC", "FDGSC", "WpeMvdrPostfilter"]
