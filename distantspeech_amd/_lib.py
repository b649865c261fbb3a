"""ctypes binding of libdsenh.so (C-ABI in include/dsenh.h).

The library holds the hand-written gfx950 kernels; there is NO CPU implementation behind this
module.  If the shared library is missing (not built) or no HIP device is visible, every entry
point fails loudly."""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# (the package does not touch the process environment.  An application that runs the SubbandGSC chain with its tail on a stream of its
# own raises the HIP runtime's hardware-queue limit itself — GPU_MAX_HW_QUEUES=8 before its first HIP call — and then says so with
# PARAM_TAIL_ASYNC: see INTEGRATION.md)
LIB_PATH = os.environ.get("DSENH_LIB", os.path.join(_HERE, "libdsenh.so"))   # DSENH_LIB: A/B of kernel builds

DS_OK = 0
ALGO_FIXED, ALGO_ADAPTIVE, ALGO_GSC = 0, 1, 2
ALGO_TRANSFORM, ALGO_MCRA, ALGO_MCMCRA, ALGO_OMLSA, ALGO_SUBLMS, ALGO_SUBRLS, ALGO_MCSPPBASE, ALGO_WPE, ALGO_MCSPP, ALGO_LINALG, ALGO_FRONTEND, ALGO_TDNLMS, ALGO_TDRLS = 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
ALGO_FDAF = 16
ALGO_ADAPTIVE_FRAMES = 17
ALGO_WPE_MVDR = 18
ALGO_SUBBAND_GSC = 19
ALGO_TDGSC = 20
ALGO_FDGSC = 21
ALGO_WPE_TD = 22
ALGO_ADAPTIVE_PF = 23
ALGO_MCSPP_MVDR = 24
PARAM_POSTFILTER = 15
PARAM_TAIL_ASYNC = 17
PARAM_REF_POWERS = 18
PARAM_WPE_FP64 = 19
PARAM_EST_POS = 20
PARAM_FDAF_TWO_PATH = 16
CHAIN_AUX_FIR, CHAIN_AUX_COHERENCE = 0, 1
PARAM_WPE_DELAY = 13
PARAM_MCSPP_REPEAT = 14
FDAF_PLAIN, FDAF_BM, FDAF_AIC = 0, 1, 2
FDAF_P_NONE, FDAF_P_BLOCK, FDAF_P_BIN = 0, 1, 2
FDAF_P_COMPLEMENT = 4
PARAM_FDAF_KIND, PARAM_FDAF_CONSTRAIN, PARAM_FDAF_NON_CAUSAL, PARAM_FDAF_WEIGHT_NORM = 9, 10, 11, 12
MEM_HOST, MEM_DEVICE = 0, 1
METHOD_SRC, METHOD_DS, METHOD_MVDR, METHOD_TFGSC = 0, 1, 2, 3
LAYOUT_SAMPLES_CHANNELS, LAYOUT_CHANNELS_SAMPLES = 0, 1
PARAM_METHOD, PARAM_MCRA_L, PARAM_ALPHA_Y, PARAM_ALPHA_V, PARAM_DIAG, PARAM_GATE, PARAM_MU, PARAM_SPLIT = 1, 2, 3, 4, 5, 6, 7, 8
(FIELD_RVV, FIELD_RYY, FIELD_MCRA_S, FIELD_MCRA_SMIN, FIELD_MCRA_STMP, FIELD_MCRA_P, FIELD_MCRA_LAMBDA_D,
 FIELD_PHI_YY, FIELD_PHI_VV, FIELD_G_AIC, FIELD_STFT_TAIL, FIELD_OLA_TAIL, FIELD_COUNTERS, FIELD_OP_STATE, FIELD_NOTCH_MEM) = range(1, 16)
FIELD_H = 16
FIELD_REF_POWERS = 17
FIELD_WPE_STATE64 = 18


class ds_config(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32), ("algo", ctypes.c_int32), ("n_mics", ctypes.c_int32),
        ("nfft", ctypes.c_int32), ("hop", ctypes.c_int32), ("batch", ctypes.c_int32),
        ("track_ryy", ctypes.c_int32), ("mcra_L", ctypes.c_int32), ("device", ctypes.c_int32),
        ("alpha_y", ctypes.c_float), ("alpha_v", ctypes.c_float), ("diag", ctypes.c_float),
        ("gate", ctypes.c_float), ("mu", ctypes.c_float),
        ("filter_len", ctypes.c_int32), ("no_norm", ctypes.c_int32), ("filt_mu", ctypes.c_float),
        ("filt_alpha", ctypes.c_float), ("rls_lambda", ctypes.c_float),
    ]


class DsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libdsenh error %d: %s" % (code, text))
        self.code = code


_lib = None

# every symbol include/dsenh.h declares (tests check that the built library exports all of them)
EXPORTS = [
    "ds_version", "ds_build_info", "ds_device_count", "ds_strerror", "ds_create", "ds_destroy", "ds_reset", "ds_last_error",
    "ds_set_steering", "ds_set_param_i", "ds_set_param_f", "ds_set_window", "ds_process", "ds_process_f64", "ds_process_pcm16", "ds_process_device",
    "ds_process_device_seq", "ds_stft", "ds_istft", "ds_mcra_estimate", "ds_mcra_estimate_p", "ds_mcmcra_estimate", "ds_mcsppbase_estimate", "ds_set_aux", "ds_mcspp_estimate", "ds_steering",
    "ds_mvdr_weight", "ds_pmwf_weight", "ds_gev_vector", "ds_blind_analytic_normalization", "ds_phase_correction", "ds_dcnotch", "ds_firbank", "ds_firbank_bm", "ds_tdfilter_update", "ds_fdaf_update", "ds_adaptive_frames", "ds_chain_set_aux", "ds_subband_gsc_process", "ds_tdgsc_process", "ds_fdgsc_process", "ds_mcspp_mvdr_process",
    "ds_omlsa_estimate", "ds_omlsa_postfilter",
    "ds_sublms_update", "ds_subrls_update", "ds_wpe_update", "ds_synchronize",
    "ds_timing_begin", "ds_timing_end", "ds_get_state", "ds_field_bytes", "ds_state_bytes", "ds_state_payload_bytes", "ds_chain_stage_info", "ds_chain_stage_field_bytes", "ds_chain_stage_state", "ds_export_state",
    "ds_import_state", "ds_host_alloc", "ds_host_free",
]


def plane_len(K):
    """lanes per plane row of the per-bin state arrays (csrc/ds_core.hpp plane_len): K bins rounded up to 8 — whole 128-byte lines per row"""
    return (int(K) + 7) & ~7


def build_info():
    """{'version': '106', 'state_layout': '4', 'arch': 'gfx950', 'shelved': '0'}: ds_build_info() parsed"""
    txt = load().ds_build_info().decode()
    return dict(kv.split("=", 1) for kv in txt.split()[1:])


def load():
    """Load libdsenh.so; raises with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "distantspeech_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C distantspeech_amd/csrc` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime; two runtimes coexist in one process only when PyTorch's initialises
    # first (observed on this image: torch 2.10+rocm7.0 next to /opt/rocm 7.2).  If torch is already imported, let it go first.
    import sys
    torch = sys.modules.get("torch")
    if torch is not None:
        try:
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cf_, cll, csz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong, ctypes.c_size_t
    lib.ds_version.restype = ci
    lib.ds_build_info.restype = ctypes.c_char_p
    lib.ds_build_info.argtypes = []
    lib.ds_device_count.restype = ci
    lib.ds_strerror.restype = ctypes.c_char_p
    lib.ds_strerror.argtypes = [ci]
    lib.ds_create.restype = ci
    lib.ds_create.argtypes = [ctypes.POINTER(ds_config), ctypes.POINTER(vp)]
    lib.ds_destroy.restype = ci
    lib.ds_destroy.argtypes = [vp]
    lib.ds_reset.restype = ci
    lib.ds_reset.argtypes = [vp]
    lib.ds_last_error.restype = ctypes.c_char_p
    lib.ds_last_error.argtypes = [vp]
    lib.ds_set_steering.restype = ci
    lib.ds_set_steering.argtypes = [vp, vp, ci]
    lib.ds_set_param_i.restype = ci
    lib.ds_set_param_i.argtypes = [vp, ci, ci]
    lib.ds_set_param_f.restype = ci
    lib.ds_set_param_f.argtypes = [vp, ci, cf_]
    lib.ds_host_alloc.restype = vp
    lib.ds_host_alloc.argtypes = [csz]
    lib.ds_host_free.restype = ci
    lib.ds_host_free.argtypes = [vp]
    lib.ds_process.restype = ci
    lib.ds_process.argtypes = [vp, vp, ci, ci, vp]
    lib.ds_process_f64.restype = ci
    lib.ds_process_f64.argtypes = [vp, vp, ci, ci, vp]
    lib.ds_process_pcm16.restype = ci
    lib.ds_process_pcm16.argtypes = [vp, vp, ci, ci, ci, vp]
    lib.ds_process_device.restype = ci
    lib.ds_process_device.argtypes = [vp, vp, ci, cll, cll, ci, vp, cll, ci, ci, vp]
    lib.ds_process_device_seq.restype = ci
    lib.ds_process_device_seq.argtypes = [vp, vp, ci, cll, cll, cll, ci, ci, vp, cll, cll, ci, ci, vp, ci]
    for name, args in (("ds_stft", [vp, vp, ci, ci, vp, ci]), ("ds_istft", [vp, vp, ci, ci, vp, ci]),
                       ("ds_mcra_estimate", [vp, vp, ci, ci, vp, ci]), ("ds_mcmcra_estimate", [vp, vp, ci, vp, vp, ci]), ("ds_mcsppbase_estimate", [vp, vp, ci, vp, vp, ci]),
                       ("ds_omlsa_estimate", [vp, vp, vp, ci, vp, vp, vp, ci]),
                       ("ds_sublms_update", [vp, vp, vp, vp, ci, vp, ci]), ("ds_subrls_update", [vp, vp, vp, ci, vp, ci]),
                       ("ds_wpe_update", [vp, vp, vp, ci, vp, ci])):
        getattr(lib, name).restype = ci
        getattr(lib, name).argtypes = args
    lib.ds_set_aux.restype = ci
    lib.ds_set_aux.argtypes = [vp, vp, csz]
    lib.ds_mcspp_estimate.restype = ci
    lib.ds_mcspp_estimate.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, ci]
    lib.ds_steering.restype = ci
    lib.ds_steering.argtypes = [vp, vp, vp, ci]
    lib.ds_mvdr_weight.restype = ci
    lib.ds_mvdr_weight.argtypes = [vp, vp, vp, vp, ci]
    lib.ds_pmwf_weight.restype = ci
    lib.ds_pmwf_weight.argtypes = [vp, vp, vp, vp, ctypes.c_float, vp, ci]
    lib.ds_gev_vector.restype = ci
    lib.ds_gev_vector.argtypes = [vp, vp, vp, vp, ci]
    lib.ds_blind_analytic_normalization.restype = ci
    lib.ds_blind_analytic_normalization.argtypes = [vp, vp, vp, ctypes.c_float, vp, ci]
    lib.ds_phase_correction.restype = ci
    lib.ds_phase_correction.argtypes = [vp, vp, vp, ci]
    lib.ds_mcra_estimate_p.restype = ci
    lib.ds_mcra_estimate_p.argtypes = [vp, vp, ci, ci, vp, vp, ci]
    lib.ds_omlsa_postfilter.restype = ci
    lib.ds_omlsa_postfilter.argtypes = [vp, vp, vp, ci, vp, vp, ci]
    lib.ds_tdfilter_update.restype = ci
    lib.ds_tdfilter_update.argtypes = [vp, vp, vp, ci, cf_, vp, ci]
    lib.ds_chain_set_aux.restype = ci
    lib.ds_chain_set_aux.argtypes = [vp, ci, vp, csz]
    lib.ds_mcspp_mvdr_process.restype = ci
    lib.ds_mcspp_mvdr_process.argtypes = [vp, vp, ci, ci, vp, vp, ci]
    lib.ds_subband_gsc_process.restype = ci
    lib.ds_subband_gsc_process.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, ci]
    lib.ds_set_window.restype = ci
    lib.ds_set_window.argtypes = [vp, vp, ci]
    lib.ds_tdgsc_process.restype = ci
    lib.ds_tdgsc_process.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, ci]
    lib.ds_fdgsc_process.restype = ci
    lib.ds_fdgsc_process.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci]
    lib.ds_adaptive_frames.restype = ci
    lib.ds_adaptive_frames.argtypes = [vp, vp, vp, ci, vp, ci]
    lib.ds_fdaf_update.restype = ci
    lib.ds_fdaf_update.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, ci]
    lib.ds_dcnotch.restype = ci
    lib.ds_dcnotch.argtypes = [vp, vp, ci, vp, ci]
    lib.ds_firbank.restype = ci
    lib.ds_firbank.argtypes = [vp, vp, ci, vp, vp, ci]
    lib.ds_firbank_bm.restype = ci
    lib.ds_firbank_bm.argtypes = [vp, vp, ci, vp, vp, vp, ci]
    lib.ds_synchronize.restype = ci
    lib.ds_synchronize.argtypes = [vp]
    lib.ds_timing_begin.restype = ci
    lib.ds_timing_begin.argtypes = [vp]
    lib.ds_timing_end.restype = ci
    lib.ds_timing_end.argtypes = [vp, ctypes.POINTER(cf_)]
    lib.ds_get_state.restype = ci
    lib.ds_get_state.argtypes = [vp, ci, vp, csz]
    lib.ds_field_bytes.restype = csz
    lib.ds_field_bytes.argtypes = [vp, ci]
    lib.ds_state_bytes.restype = csz
    lib.ds_state_bytes.argtypes = [vp]
    lib.ds_state_payload_bytes.restype = csz
    lib.ds_state_payload_bytes.argtypes = [vp]
    lib.ds_chain_stage_field_bytes.restype = csz
    lib.ds_chain_stage_field_bytes.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    lib.ds_chain_stage_state.restype = ctypes.c_int
    lib.ds_chain_stage_state.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, csz]
    lib.ds_chain_stage_info.restype = ctypes.c_int
    lib.ds_chain_stage_info.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(csz)]
    lib.ds_export_state.restype = ci
    lib.ds_export_state.argtypes = [vp, vp, csz]
    lib.ds_import_state.restype = ci
    lib.ds_import_state.argtypes = [vp, vp, csz]
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc != DS_OK:
        lib = load()
        text = lib.ds_last_error(handle)
        text = text.decode() if text else lib.ds_strerror(rc).decode()
        raise DsError(rc, text)
