/* dsenh.h — C-ABI of libdsenh.so: MI355X-native per-frame multichannel speech enhancement
 * (streaming STFT -> beamformer -> post-filter -> ISTFT overlap-add) batched over independent
 * utterances.  Plain C, no torch / HIP types in any signature (streams travel as void*).
 *
 * Each entry point replaces a piece of the reference's Python object interface
 * (wangwei2009/DistantSpeech; paths relative to the reference root):
 *
 *   ds_create / ds_destroy / ds_reset
 *       constructors + attribute state of  beamformer/beamformer.py:223-265,
 *       beamformer/fixedbeamformer.py:96-107, beamformer/adaptivebeamformer.py:10-42,
 *       beamformer/GSC.py:27-87, transform/transform.py:407-428 (state: previous_input/output),
 *       noise_estimation/NoiseEstimationBase.py:5-31, noise_estimation/mc_mcra.py:25-80
 *   ds_set_steering
 *       the per-look-direction set-up the reference redoes inside process():
 *       a[k,m] = exp(-j w_k tau_m)  (adaptivebeamformer.py:52,84 ; GSC.py:186,205-222) or the fixed
 *       weights W[k,m] (beamformer.py:338-373 ; fixedbeamformer.py:109-145)
 *   ds_process / ds_process_device
 *       FixedBeamformer.process        fixedbeamformer.py:167-207   (algo DS_ALGO_FIXED)
 *       adaptivebeamfomer.process      adaptivebeamformer.py:44-128 (algo DS_ALGO_ADAPTIVE)
 *       GSC.process                    GSC.py:174-294               (algo DS_ALGO_GSC)
 *       each including Transform.stft / Transform.istft (transform.py:430-481) and the realtime
 *       callback contract  realtime/realtime_processing.py:78-84 (chunk in -> chunk out, state carried)
 *   ds_get_state
 *       the attributes users read back after process(): Rvv/Ryy (adaptivebeamformer.py:32-34),
 *       mcra.S/Smin/Stmp/p/lambda_d (NoiseEstimationBase.py:11-24), spp.Phi_yy/Phi_vv
 *       (mc_mcra.py:68-69), G (GSC.py:72), Transform.previous_input/previous_output (transform.py:425-426)
 *   ds_set_param_i / ds_set_param_f
 *       attributes users poke after construction (e.g. mcra.L = 10 in the notebooks; `method` argument)
 *
 * Conventions: every function returns 0 (DS_OK) or a negative DS_E* code; ds_last_error() gives
 * the text.  A handle is NOT thread-safe (the reference objects are single-caller too).
 * Chunking contract: n_samples must be a multiple of hop; state is carried across calls; a call
 * with T hops is defined as T successive one-hop calls of the reference.
 */
#ifndef DSENH_H
#define DSENH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_VERSION 108   /* 108 (round 6): ds_process_f64, ds_host_alloc / ds_host_free, DS_PARAM_EST_POS, DS_ALGO_ADAPTIVE_PF at 8 microphones */
#define DS_STATE_LAYOUT 5   /* 5 (round 5): the DC notch memories of a front-end handle are doubles ([B][M][2] x 8 bytes in a checkpoint; DS_FIELD_NOTCH_MEM
                               reads them back as float32).  Before that, 4: serialised arrangement of the carried state (checkpoint header; measurement records quote it).  4 (round 4): plane rows of
                               KP = K rounded up to 8 lanes (whole 128-byte lines per row of float4 words), RLS-WPE blocks on 128-byte lines (csrc/ds_wpe.hpp
                               wpe_layout); ds_import_state refuses blobs of another layout */

/* error codes */
#define DS_OK 0
#define DS_EINVAL (-1)       /* bad argument / NULL handle                       */
#define DS_ESHAPE (-2)       /* n_samples not a multiple of hop, size mismatch    */
#define DS_EUNSUPPORTED (-3) /* configuration with no compiled kernel             */
#define DS_EHIP (-4)         /* HIP runtime error (text in ds_last_error)         */
#define DS_ENOMEM (-5)
#define DS_ESTATE (-6)       /* call order (e.g. process before set_steering)     */

/* ds_config.algo */
#define DS_ALGO_FIXED 0      /* FixedBeamformer: Y = sum conj(W) X                */
#define DS_ALGO_ADAPTIVE 1   /* adaptivebeamfomer: MCRA-gated Rvv, src/DS/MVDR/TFGSC */
#define DS_ALGO_GSC 2        /* GSC: FD blocking matrix + SPP-controlled LMS + McMcra gain */
/* frame-level objects (drive them with the ds_stft / ds_*_estimate / ds_sub*_update entry points) */
#define DS_ALGO_TRANSFORM 3  /* Transform.stft / istft               transform/transform.py:407-481 */
#define DS_ALGO_MCRA 4       /* NoiseEstimationMCRA.estimation       noise_estimation/mcra.py:27-77 */
#define DS_ALGO_MCMCRA 5     /* McMcra.estimation                    noise_estimation/mc_mcra.py:179-224 */
#define DS_ALGO_OMLSA 6      /* NsOmlsaMulti.estimation              noise_estimation/omlsa_multi.py:73-156 */
#define DS_ALGO_SUBLMS 7     /* SubbandLMS (n_mics=1) / SubbandLmsMc (n_mics=C) .update   adaptivefilter/SubbandLMS.py, SubbandLmsMc.py */
#define DS_ALGO_SUBRLS 8     /* SubbandRLS.update                    adaptivefilter/SubbandRLS.py:44-71 */
#define DS_ALGO_MCSPPBASE 9  /* McSppBase.estimation + PMWF weights  noise_estimation/mcspp_base.py:220-324 */
#define DS_ALGO_WPE 10       /* Wpe.update frequency-domain core (RLS-WPE on the STFT grid)  dereverberation/awpe.py:129-192; n_mics <= 8, n_mics * filter_len <= 80 */
#define DS_ALGO_MCSPP 11     /* McSpp.estimation (McCDR prior) + fused steering/MVDR   noise_estimation/mcspp.py:244-305, mccdr.py:122-177 */
#define DS_ALGO_LINALG 12    /* stateless per-bin helpers: steering(), compute_mvdr_weight()   beamformer/beamformer.py:10-31,133-155 */
#define DS_ALGO_FRONTEND 13  /* time-domain conditioning: FilterDcNotch16 (feature.py:32-49), TimeAlignment FIR bank (fixedbeamformer.py:13-93) */
#define DS_ALGO_TDNLMS 14     /* BaseFilter.update (sample-wise NLMS)  adaptivefilter/BaseFilter.py:52-85; filter_len <= 1024 */
#define DS_ALGO_TDRLS 15      /* Rls.update (sample-wise RLS)         adaptivefilter/RLS.py:26-42; filter_len <= 256 (P in LDS up to 64 taps, in device memory beyond) */
#define DS_ALGO_FDAF 16       /* overlap-save FDAF block filters: FastFreqLms.update (adaptivefilter/FastFreqLms.py:204-245),
                                 AdaptiveBlockingMatrixFilter.update (beamformer/gsc_bm.py:61-122), AdaptiveInterferenceCancellation.update
                                 (beamformer/gsc_aic.py:53-108); nfft = 2 * filter_len in {128,256,512,1024}, n_mics = input channels <= 8 */
#define DS_ALGO_ADAPTIVE_FRAMES 17 /* adaptivebeamfomer's frame loop on STFT frames (adaptivebeamformer.py:69-120): the per-bin program of
                                     DS_ALGO_ADAPTIVE as a frame-level operator, with an optional post-filter gain input */
#define DS_ALGO_WPE_MVDR 18    /* BASELINE config 4 as ONE handle behind ds_process / ds_process_device: STFT -> RLS-WPE on all channels
                                 (awpe.py:152-189; ds_config.filter_len taps, rls_lambda forgetting factor, prediction delay DS_PARAM_WPE_DELAY
                                 frames, default 4) -> McMcra gain (mc_mcra.py:179-224) -> adaptive MVDR frame loop (adaptivebeamformer.py:69-120)
                                 x gain -> ISTFT; device-resident between the stages, all on the handle's stream; n_mics * filter_len <= 80
                                 (16 < n_mics * filter_len: the wavefront-per-bin kernel, csrc/ds_wpe_wide.hpp) */
#define DS_ALGO_SUBBAND_GSC 19 /* SubbandGSC.process (beamformer/SubbandGSC.py:170-262; BASELINE config 5 with rls_lambda > 0) as ONE handle:
                                 DC notch -> TimeAlignment FIR bank + mean beamformer -> STFT -> McSpp -> M adaptive blocking filters (one
                                 batched subband LMS, or RLS when ds_config.rls_lambda > 0) -> ISTFT/STFT -> multichannel subband-LMS canceller
                                 -> ISTFT, device-resident between the stages; nfft = 2 * frameLen, hop = frameLen, n_mics in {2,4,6},
                                 filter_len taps (0 -> 2).  Needs ds_chain_set_aux() for the FIR bank and the McCDR coherence first */
#define DS_ALGO_TDGSC 20       /* TDGSC.process (beamformer/TDGSC.py:110-175) as ONE handle: DC notch -> TimeAlignment FIR bank + channel mean +
                                  pairwise-difference blocking matrix -> STFT -> MCRA (L = 65) -> multichannel overlap-save canceller
                                  (FastFreqLms, non-causal, p = 1 - p, fir_truncate 30) -> optional OMLSA post-filter; ds_tdgsc_process,
                                  ds_process_device (post-filter per DS_PARAM_POSTFILTER); FIR table via ds_chain_set_aux(DS_CHAIN_AUX_FIR) */
#define DS_ALGO_FDGSC 21       /* FDGSC.process (beamformer/FDGSC.py:201-317, blocking-matrix mode 3) as ONE handle: DC notch -> FIR bank + mean ->
                                  MCRA (L = 60) on channel 0 + the :248-255 adaptation control -> M clamped adaptive blocking filters (one
                                  batched launch, shared fixed-beamformer input, half-block-delayed aligned channels as desired signals) ->
                                  norm-limited multichannel canceller on the one-block-delayed fixed output -> optional OMLSA post-filter
                                  (block-sequential like the reference: its two signals share one Transform); ds_fdgsc_process */

#define DS_ALGO_WPE_TD 22      /* Wpe.update (dereverberation/awpe.py:129-192) as ONE call behind ds_process / ds_process_device: analysis of the
                                  n_mics channels -> prediction delay line (DS_PARAM_WPE_DELAY frames, default 4: DelaySamples of delay * hop
                                  samples, awpe.py:75-76,147) -> RLS-WPE on all channels (filter_len taps, rls_lambda forgetting factor) ->
                                  synthesis of the dereverberated channel 0, all on the device.  hop = nfft / 2 or nfft / 4 (the reference's
                                  maintained use is Wpe(channels=4, filter_len=20, delay=4, num_bands=256, hop_length=64), example/wpe.ipynb
                                  cell 2); n_mics <= 8, n_mics * filter_len <= 80.  No steering vector.  State of the filter: the stage handle
                                  of ds_chain_stage_info stage 1 (DS_FIELD_OP_STATE: per-bin blocks, csrc/ds_wpe.hpp) */

/* `mem` argument of the frame-level entry points */
#define DS_MEM_HOST 0
#define DS_MEM_DEVICE 1

/* `method` (AlgorithmList, adaptivebeamformer.py:36) */
#define DS_METHOD_SRC 0
#define DS_METHOD_DS 1
#define DS_METHOD_MVDR 2
#define DS_METHOD_TFGSC 3    /* needs ds_config.track_ryy = 1 */

/* sample layouts of x (per utterance) */
#define DS_LAYOUT_SAMPLES_CHANNELS 0   /* x[L][M]  (FixedBeamformer.process, Transform.stft) */
#define DS_LAYOUT_CHANNELS_SAMPLES 1   /* x[M][L]  (adaptivebeamfomer.process, GSC.process)  */

typedef struct ds_handle ds_handle;

typedef struct ds_config {
    int32_t struct_size; /* = sizeof(ds_config) */
    int32_t algo;
    int32_t n_mics;      /* M: 2..6 or 8 for the beamformer objects (7 has no array geometry in MicArray.py), 1..8 for Transform */
    int32_t nfft;        /* 256, 512 or 1024 */
    int32_t hop;         /* nfft / 2 (the only overlap the reference's beamformer callers use); DS_ALGO_TRANSFORM also takes nfft / 4 */
    int32_t batch;       /* independent utterances resident on this device */
    int32_t track_ryy;   /* adaptive: also keep Ryy like the reference (needed by TFGSC) */
    int32_t mcra_L;      /* 0 -> 15 (mcra.py:25) */
    int32_t device;      /* HIP ordinal, -1 = current device */
    float alpha_y;       /* 0 -> 0.8     adaptivebeamformer.py:65 */
    float alpha_v;       /* 0 -> 0.9998  adaptivebeamformer.py:66 */
    float diag;          /* 0 -> 1e-6    adaptivebeamformer.py:89 */
    float gate;          /* 0 -> 0.4     adaptivebeamformer.py:94 */
    float mu;            /* 0 -> 0.01    GSC.py:202 */
    /* frame-level filters (DS_ALGO_SUBLMS / DS_ALGO_SUBRLS); 0 -> the reference's defaults */
    int32_t filter_len;  /* taps per band, 1..4; 0 -> 2           SubbandAF.py:15 */
    int32_t no_norm;     /* 1 -> plain LMS (normalization=False)  SubbandAF.py:18 */
    float filt_mu;       /* 0 -> 0.1 (LMS) / 0.5 (RLS)            SubbandAF.py:17, SubbandRLS.py:17 */
    float filt_alpha;    /* 0 -> 0.9   power smoothing            SubbandAF.py:19 */
    float rls_lambda;    /* 0 -> 0.998 forgetting factor          SubbandRLS.py:16 */
} ds_config;

#define DS_ALGO_ADAPTIVE_PF 23  /* adaptivebeamfomer.process (adaptivebeamformer.py:69-120) with the McMcra speech-presence gain of the same input frame
                                  applied to its output, Y = w^H z * spp.G — the post-filter convention of GSC.py:225,286 (spp = McMcra,
                                  mc_mcra.py:179-224) — in ONE fused frame kernel behind ds_process / ds_process_device: "MVDR + post-filter" in
                                  one pass.  Methods src / DS / MVDR; parameters and fields of DS_ALGO_ADAPTIVE (track_ryy = 0) plus
                                  DS_FIELD_PHI_YY / DS_FIELD_PHI_VV; n_mics 2..6 or 8 (round 6: 8 microphones and 6 at 1024 points too — at 8 the two
                                  programs hold 141 state floats per lane: one wave per SIMD, and the 1024-point kernel spills) */

#define DS_ALGO_MCSPP_MVDR 24   /* the online MVDR of example/mvdr.ipynb cell 4 as ONE handle behind ds_process / ds_process_device / ds_mcspp_mvdr_process:
                                  Transform.stft of the n_mics channels (transform.py:430-453) -> per frame McSpp.estimation (McCDR prior; mcspp.py:244-305,
                                  mccdr.py:122-177) -> steering(Phi_xx) (beamformer.py:10-31) -> compute_mvdr_weight(steer, Phi_vv_inv)
                                  (beamformer.py:133-155) -> Y = w^H y -> Transform.istft (transform.py:455-481).  Stages (ds_chain_stage_info):
                                  0 analysis transform, 1 DS_ALGO_MCSPP, 2 synthesis transform.  Diffuse coherence via
                                  ds_chain_set_aux(DS_CHAIN_AUX_COHERENCE) first; DS_PARAM_MCSPP_REPEAT as for DS_ALGO_MCSPP; n_mics 4 or 6
                                  (8: the reference's own McSpp raises LinAlgError on its 8-channel recording — inv(Phi_yy) of a rank-deficient matrix, mcspp.py:226 —
                                  so there is nothing to be identical to; tests/golden/make_golden.py g11b) */

/* ds_set_param_* ids */
#define DS_FDAF_PLAIN 0
#define DS_FDAF_BM 1
#define DS_FDAF_AIC 2
#define DS_FDAF_P_NONE 0
#define DS_FDAF_P_BLOCK 1
#define DS_FDAF_P_BIN 2
#define DS_FDAF_P_COMPLEMENT 4   /* OR-ed into p_mode: the kernel uses 1 - p (TDGSC.py:154) */
#define DS_PARAM_METHOD 1   /* int   */
#define DS_PARAM_MCRA_L 2   /* int   */
#define DS_PARAM_ALPHA_Y 3  /* float */
#define DS_PARAM_ALPHA_V 4
#define DS_PARAM_DIAG 5
#define DS_PARAM_GATE 6
#define DS_PARAM_MU 7
#define DS_PARAM_FDAF_KIND 9         /* int: DS_FDAF_PLAIN / DS_FDAF_BM / DS_FDAF_AIC (DS_ALGO_FDAF handles) */
#define DS_PARAM_FDAF_CONSTRAIN 10   /* int 0/1: gradient (plain) or coefficient (bm, aic) constraint; default 1 */
#define DS_PARAM_FDAF_NON_CAUSAL 11  /* int 0/1: delay the desired signal by filter_len / 2 (FastFreqLms.py:84-85,167-168); default 0 */
#define DS_PARAM_FDAF_WEIGHT_NORM 12 /* int 0/1: norm limiter of the canceller (gsc_aic.py:81-88); default 0 */
#define DS_PARAM_WPE_DELAY 13        /* int >= 0: prediction delay of the DS_ALGO_WPE_MVDR chain in frames (awpe.py:36, default 4); set before the first call */
#define DS_PARAM_MCSPP_REPEAT 14     /* int 0/1: DS_ALGO_MCSPP handles run estimation(repeat=True): a second estimation_core after the noise update (mcspp.py:280-282); default 0 */
#define DS_PARAM_FDAF_TWO_PATH 16     /* int 0/1: FastFreqLms(two_path=True) — foreground / background filters with the 3 dB transfer rule
                                         (FastFreqLms.py:94-104,162-176); plain kind only; default 0 */
#define DS_PARAM_TAIL_ASYNC 17        /* int 0/1, DS_ALGO_SUBBAND_GSC, before the first call: run the chain's tail kernel on a stream of its own.  Pays only
                                         when the process's HIP runtime has >= 6 hardware queues per device (the APPLICATION sets GPU_MAX_HW_QUEUES=8
                                         before its first HIP call; the runtime's default is 4 and the library cannot query it); default 0, or
                                         DS_CHAIN_TAIL_ASYNC=1 in the environment at ds_create */
#define DS_PARAM_REF_POWERS 18       /* int 0/1, DS_ALGO_GSC: every ds_process / ds_process_device call also keeps, per frame and bin, the power of the canceller
                                         output in front of the post-filter gain and the powers of the M - 1 blocking-matrix outputs — the arguments of
                                         omlsa_multi.estimation in GSC.process (GSC.py:281-283; the reference computes that estimate and uses nothing of
                                         it).  Read with ds_get_state(DS_FIELD_REF_POWERS) after the call.  One plain call at a time: sequences of several
                                         calls, hipGraph replays and utterance groups are refused while it is on; default 0 */
#define DS_PARAM_WPE_FP64 19         /* int 0/1, before the first frame, DS_ALGO_WPE / DS_ALGO_WPE_TD / DS_ALGO_WPE_MVDR: run the RLS-WPE recursion (P, W, taps, var) in
                                         double like the reference's complex128 (awpe.py:60-71,181-186) instead of fp32 — one workgroup per (utterance,
                                         bin), several times slower and four times the state bytes: for streams where the fp32 recursion's
                                         eps x cond(P) matters (csrc/ds_wpe64.hpp).  State read-back: DS_FIELD_WPE_STATE64 */
#define DS_PARAM_EST_POS 20          /* int, DS_ALGO_ADAPTIVE / DS_ALGO_ADAPTIVE_PF: adaptivebeamfomer.estPos (adaptivebeamformer.py:30,90-93).  -1 (default) = None:
                                         the MCRA-based gate (:94).  n >= 0: the noise covariance Rvv is updated for the first n (frame, bin) slots
                                         after a reset and never afterwards — the reference's frameCount advances once per BIN, so n = 30 * (nfft / 2 + 1)
                                         is "the first 30 frames", and the one frame in which the count runs out updates its leading bins only.  The
                                         count restarts at ds_reset, at every ds_set_steering (the reference restarts it when the look direction
                                         changes, :70-79) and when DS_PARAM_METHOD changes.  While n >= 0 a handle processes its whole batch per call,
                                         with plain launches (no hipGraph replay, no utterance groups) */
#define DS_PARAM_POSTFILTER 15       /* int 0/1: DS_ALGO_TDGSC / DS_ALGO_FDGSC handles apply the OMLSA post-filter in ds_process_device; default 0 */
#define DS_PARAM_SPLIT 8   /* int 1..8: utterance groups.  Fused frame kernels: ds_process_device_seq runs the utterance range as that many groups,
                              each on its own stream at its own pace (its own hipGraph with graph=1); default 2 from 2048 utterances up, else 1.
                              DS_ALGO_WPE_MVDR: the batch as that many independent chains on their own streams; default 2 from 512 utterances up.
                              Results are complete after ds_synchronize (every other entry point joins the groups first) */

/* ds_get_state fields; all arrays are float32, complex = interleaved (re, im) */
#define DS_FIELD_RVV 1        /* [B][K][M][M][2]  */
#define DS_FIELD_RYY 2        /* [B][K][M][M][2]  (track_ryy) */
#define DS_FIELD_MCRA_S 3     /* [B][K] */
#define DS_FIELD_MCRA_SMIN 4
#define DS_FIELD_MCRA_STMP 5
#define DS_FIELD_MCRA_P 6
#define DS_FIELD_MCRA_LAMBDA_D 7
#define DS_FIELD_PHI_YY 8     /* [B][K][M][M] real */
#define DS_FIELD_PHI_VV 9
#define DS_FIELD_G_AIC 10     /* [B][K][M-1][2] */
#define DS_FIELD_STFT_TAIL 11 /* [B][M][nfft - hop]  Transform.previous_input (transform.py:424-425)  */
#define DS_FIELD_OLA_TAIL 12  /* [B][hop] of a beamformer object; [B][M][nfft - hop] of a DS_ALGO_TRANSFORM handle: Transform.previous_output */
#define DS_FIELD_COUNTERS 13  /* int32 [B][4] {mcra.frm_cnt, mcra.ell, spp.frm_cnt, 0} */
#define DS_FIELD_OP_STATE 14  /* frame-level objects: raw state [B][NF / 4][KP][4] float32, KP = K rounded up to 8 (row map in distantspeech_amd/ops.py;
                                 RLS-WPE objects: one block per (utterance, bin), distantspeech_amd/ops.py wpe_block_layout) */
#define DS_FIELD_H 16          /* DS_ALGO_ADAPTIVE, methods src / DS / MVDR: [B][K][M][2] the weights the frame kernel applies to the next frame
                                 (adaptivebeamformer.py:105-112: H[:, k]), computed by the kernel's own fused Cholesky solve on the handle's Rvv —
                                 a read-only probe; needs ds_set_steering */
#define DS_FIELD_REF_POWERS 17 /* DS_ALGO_GSC with DS_PARAM_REF_POWERS: [B][T][K][M] of the LAST call (T = its hops): [0] = |Y|^2, [1 + i] = |U_i|^2 */
#define DS_FIELD_WPE_STATE64 18 /* DS_ALGO_WPE with DS_PARAM_WPE_FP64: float64 [B][K][2 CN CN + 2 C CN + 2 CN + 2]: P row-major, W [C][CN], taps, var, pad (csrc/ds_wpe64.hpp) */
#define DS_FIELD_NOTCH_MEM 15 /* DS_ALGO_FRONTEND: [B][M][2] the DC notch memories (FilterDcNotch16.notch_mem, feature.py:34,47) */

int ds_version(void);
/* one line identifying the build for measurement records and tests: version, state layout, optional kernel sets compiled in
   ("shelved=1": the hop-pipelined and quad-lane experiment kernels, make SHELVED=1) */
const char* ds_build_info(void);
int ds_device_count(void);
const char* ds_strerror(int code);

int ds_create(const ds_config* cfg, ds_handle** out);
int ds_destroy(ds_handle* h);
/* zero all per-utterance state (like constructing fresh reference objects) */
int ds_reset(ds_handle* h);
const char* ds_last_error(const ds_handle* h);

/* steer: complex float [K][M] (per_utterance == 0) or [B][K][M] (per_utterance != 0), host memory.
 * For DS_ALGO_FIXED these are the beamformer weights W; otherwise the steering vector a. */
int ds_set_steering(ds_handle* h, const float* steer, int per_utterance);

int ds_set_param_i(ds_handle* h, int id, int value);
/* DS_ALGO_TRANSFORM handles: analysis / synthesis window of n = nfft samples instead of the default sqrt-Hann (Transform(window=...),
 * transform/transform.py:415-419); the synthesis scale hop / sum(window^2) (:428,479) follows */
int ds_set_window(ds_handle* h, const float* window, int n);
int ds_set_param_f(ds_handle* h, int id, float value);

/* Host-buffer call (the realtime-callback form): x is [B] x layout, y is [B][n_samples];
 * synchronous: returns after the enhanced samples are in y. */
int ds_process(ds_handle* h, const float* x, int layout, int n_samples, float* y);
/* The same call with the enhanced samples handed back as float64 [B][n_samples] — what the reference's process() returns (transform.py:479) —
 * widened on the device (exact): on the host that conversion is the longest part of a call at batch */
int ds_process_f64(ds_handle* h, const float* x, int layout, int n_samples, double* y);

/* Realtime wire format (realtime/realtime_processing.py:113-136): pcm = int16 little-endian interleaved [B][n_samples][n_total_channels]
 * straight from the capture device; channels [first_channel, first_channel + n_mics) are the microphones; they are scaled by
 * 1/32768 exactly like the shell does (:119), enhanced, and written back as int16 (y * 32768, truncated; out-of-range saturates)
 * to out [B][n_samples].  Conversion happens on the GPU. */
int ds_process_pcm16(ds_handle* h, const int16_t* pcm, int n_total_channels, int first_channel, int n_samples, int16_t* out);

/* Device-buffer call: pointers are HIP device memory, 16-byte aligned; strides in elements.
 * x_chan_stride: elements between channels for DS_LAYOUT_CHANNELS_SAMPLES (0 = n_samples, i.e. a
 * dense [M][n_samples] chunk; pass the row length when x is a window into a longer [M][L] recording).
 * Asynchronous on `stream` (a hipStream_t passed as void*; NULL = the handle's own stream).
 * Processes utterances [first, first + count) of the handle; x/y point at utterance `first`. */
int ds_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                      int n_samples, float* y_dev, long long y_batch_stride, int first, int count, void* stream);

/* n_calls successive ds_process_device() calls enqueued from one host call: call i reads
 * x_dev + i * x_call_stride and writes y_dev + i * y_call_stride (elements).  This is the realtime
 * shell's loop (realtime/realtime_processing.py:113-136: one process() per captured chunk) hoisted into
 * the library so the host does not pay one FFI crossing per chunk.
 * graph: 0 = plain launches; 1 = replay the sequence as a hipGraph (captured and cached on first use for
 * this exact argument set); 2 = only build/cache the graph, launch nothing. */
int ds_process_device_seq(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                          long long x_call_stride, int n_samples_per_call, int n_calls, float* y_dev,
                          long long y_batch_stride, long long y_call_stride, int first, int count, void* stream,
                          int graph);

/* ---- frame-level entry points: the L2 objects callers drive frame by frame.
 * Arrays carry a leading batch dimension B and a frame dimension T ("T successive calls"); complex =
 * interleaved (re, im) float32; `mem` = DS_MEM_HOST (synchronous, staged) or DS_MEM_DEVICE (asynchronous
 * on the handle's stream).
 *   ds_stft            x [B] x layout, n_samples  -> Y complex [B][T][K][M]            (Transform.stft)
 *   ds_istft           Y complex [B][T][K][C], C <= M -> y [B][T*hop][C]              (Transform.istft)
 *   ds_mcra_estimate   Y [B][T][K] power (or complex if is_complex) -> lambda_d [B][T][K]
 *   ds_mcmcra_estimate y complex [B][T][K][M] -> p, G [B][T][K]
 *   ds_mcsppbase_estimate y complex [B][T][K][M] -> p [B][T][K], w complex [B][T][K][M] (PMWF weights)
 *   ds_mcspp_estimate  y complex [B][T][K][M] -> p [B][T][K]; optional (NULL to skip): w_pmwf complex [B][T][K][M],
 *                      yout complex [B][T][K] = the notebook's online MVDR output (steering(Phi_xx) -> compute_mvdr_weight
 *                      -> sum conj(w) y, example/mvdr.ipynb cell 4), phi_xx / phi_vv_inv complex [B][T][K][M][M].
 *                      Needs ds_set_aux(h, Fn[K]) first: diffuse coherence of microphones 1,2 (mccdr.py:141).
 *   ds_steering        XX complex [B][K][M][M] -> v complex [B][K][M]     (DS_ALGO_LINALG handle)
 *   ds_mvdr_weight     steer complex [B][K][M], Rinv complex [B][K][M][M] -> w complex [B][K][M]
 *   ds_dcnotch         x [B][M][n] -> y [B][M][n]   (DS_ALGO_FRONTEND handle; radius = ds_config.filt_alpha, 0 -> 0.9)
 *   ds_firbank         x [B][n][M] -> y [B][n][M] (+ optional channel mean [B][n]); coefficients [L][M] via ds_set_aux
 *   ds_firbank_bm      ds_firbank + optional bm [B][n][M-1] = y[m] - y[m+1], the fixed blocking matrix of TDGSC (TDGSC.py:69-87)
 *   ds_tdfilter_update x [B][n], d [B][n] samples -> err [B][n]; n successive BaseFilter.update / Rls.update calls
 *                      (DS_ALGO_TDNLMS: filt_mu 0 -> 0.1, p = update probability; DS_ALGO_TDRLS: filt_mu 0 -> 0.5,
 *                      rls_lambda 0 -> 0.9998); weights via ds_get_state(DS_FIELD_OP_STATE) = [B][L]
 *   ds_adaptive_frames Z complex [B][T][K][M], gain [B][T][K] or NULL -> Y complex [B][T][K] = (w^H Z) * gain with the MCRA-gated
 *                      Rvv recursion and src/DS/MVDR weights of adaptivebeamfomer.process (DS_ALGO_ADAPTIVE_FRAMES handle;
 *                      ds_set_steering, DS_PARAM_METHOD / MCRA_L / ALPHA_V / DIAG / GATE as for DS_ALGO_ADAPTIVE)
 *   ds_fdaf_update     x [B][T*L][C], d [B][T*L] samples (L = nfft / 2 = filter_len) -> err [B][T*L]; T successive .update calls of
 *                      the kind selected with DS_PARAM_FDAF_KIND (mu = filt_mu, 0 -> 0.1; alpha = filt_alpha, 0 -> 0.9).
 *                      p_mode DS_FDAF_P_NONE (p = 1), DS_FDAF_P_BLOCK (p [B][T]) or DS_FDAF_P_BIN (p [B][T][K]);
 *                      fir_truncate < 0 = None; w_out (or NULL) [B][L][C] = self.w after the last block.
 *                      State (W, P, input and delay buffers) via ds_get_state(DS_FIELD_OP_STATE), see DESIGN.md.
 *   ds_omlsa_estimate  y [B][T][K], u [B][T][K][M-1] powers -> lambda_d, G, p [B][T][K]
 *   ds_omlsa_postfilter Y complex [B][T][K], U complex [B][T][K][M-1] -> G [B][T][K], Yout = Y * sqrt(G) complex [B][T][K]: the
 *                      post-filter step of TDGSC / FDGSC (TDGSC.py:158-168) with the powers and the gain applied in the kernel
 *   ds_sublms_update   x complex [B][T][K][C], d complex [B][T][K], p [B][T][K] or NULL -> err complex [B][T][K]
 *   ds_subrls_update   x complex [B][T][K], d complex [B][T][K] -> err complex [B][T][K]
 *   ds_wpe_update      x_delayed complex [B][T][K][C], d complex [B][T][K][C] -> err complex [B][T][K][C]
 *                      (n_mics = C channels, filter_len = taps; the caller supplies the `delay`-hops-old frame)    */
int ds_stft(ds_handle* h, const float* x, int layout, int n_samples, float* Y, int mem);
int ds_istft(ds_handle* h, const float* Y, int n_frames, int n_channels, float* y, int mem);
int ds_mcra_estimate(ds_handle* h, const float* Y, int is_complex, int n_frames, float* lambda_d, int mem);
int ds_mcra_estimate_p(ds_handle* h, const float* Y, int is_complex, int n_frames, float* lambda_d, float* p, int mem);   /* + p [B][T][K] (mcra.p after each frame) */
int ds_mcmcra_estimate(ds_handle* h, const float* y, int n_frames, float* p, float* G, int mem);
int ds_mcsppbase_estimate(ds_handle* h, const float* y, int n_frames, float* p, float* w, int mem);
int ds_set_aux(ds_handle* h, const float* table, size_t n_floats);
int ds_mcspp_estimate(ds_handle* h, const float* y, int n_frames, float* p, float* w_pmwf, float* yout, float* phi_xx,
                      float* phi_vv_inv, int mem);
int ds_steering(ds_handle* h, const float* XX, float* v, int mem);
int ds_mvdr_weight(ds_handle* h, const float* steer, const float* Rinv, float* w, int mem);
/* The other free functions of beamformer/beamformer.py the notebook flows use (mvdr.ipynb: get_gev_vector -> phase_correction ->
 * blind_analytic_normalization; beamformer.py:34-130), on a DS_ALGO_LINALG handle, double arithmetic inside, complex64 in / out:
 *   ds_pmwf_weight                    xi [B][K], Rxx / Rvv_inv complex [B][K][M][M], beta -> w complex [B][K][M] = (Rvv_inv Rxx)[:, 0] / (beta + xi)
 *                                     (beamformer.py:100-130; its `channels = Rxx.shape[0]` only works for M == bins: [bins, M, M] matrices here)
 *   ds_gev_vector                     target / noise complex [B][K][M][M] -> principal generalised eigenvector [B][K][M], v^H N v = 1
 *                                     (beamformer.py:79-97 = scipy.linalg.eigh(a, b)[1][:, -1]; phase: first whitened component real positive)
 *   ds_blind_analytic_normalization   vector [B][K][M], noise [B][K][M][M], eps -> vector * |sqrt(v^H N N v)| / (|v^H N v| + eps)   (:34-63)
 *   ds_phase_correction               vector [B][K][M] -> bin f rotated onto bin f - 1 (serial over the bins of an utterance)      (:66-76) */
int ds_pmwf_weight(ds_handle* h, const float* xi, const float* Rxx, const float* Rvv_inv, float beta, float* w, int mem);
int ds_gev_vector(ds_handle* h, const float* target, const float* noise, float* v, int mem);
int ds_blind_analytic_normalization(ds_handle* h, const float* vector, const float* noise, float eps, float* out, int mem);
int ds_phase_correction(ds_handle* h, const float* vector, float* out, int mem);
int ds_dcnotch(ds_handle* h, const float* x, int n_samples, float* y, int mem);
int ds_firbank(ds_handle* h, const float* x, int n_samples, float* y, float* mean, int mem);
int ds_firbank_bm(ds_handle* h, const float* x, int n_samples, float* y, float* mean, float* bm, int mem);
int ds_tdfilter_update(ds_handle* h, const float* x, const float* d, int n_samples, float p, float* err, int mem);
#define DS_CHAIN_AUX_FIR 0        /* TimeAlignment coefficients [L][M] (fixedbeamformer.py:68-70) */
#define DS_CHAIN_AUX_COHERENCE 1 /* diffuse coherence Fn[K] of microphones 1, 2 (mccdr.py:141) */
int ds_chain_set_aux(ds_handle* h, int which, const float* table, size_t n_floats);
/* DS_ALGO_SUBBAND_GSC: x [B][M][n] (n a multiple of hop) -> y [B][n]; optional (NULL to skip) fix_output [B][n] (the fixed beamformer
 * output delayed by one block), bm_output [B][M][n], p [B][T][K], aligned [B][M][n] — the tuple SubbandGSC.process returns */
int ds_subband_gsc_process(ds_handle* h, const float* x, int n_samples, float* y, float* fix_output, float* bm_output, float* p,
                           float* aligned, int mem);
/* DS_ALGO_TDGSC: x [B][M][n] channel-major (n a multiple of frameLen = hop) -> out [B][n]; optional (NULL to skip) p [B][T][K] (MCRA speech
 * presence probability per block), bm [B][n][M-1] (blocking-matrix outputs), w [B][frameLen][M-1] (canceller coefficients after the
 * last block) — TDGSC.process's tuple (+ aic_filter.w) */
int ds_tdgsc_process(ds_handle* h, const float* x, int n_samples, int postfilter, float* out, float* p, float* bm, float* w, int mem);
/* DS_ALGO_FDGSC: x [B][M][n] channel-major -> out [B][n]; optional p [B][T][K], fix_output [B][n], fix_output_delayed [B][n], bm_output
 * [B][M][n], aligned [B][M][n], aligned_delayed [B][M][n] (channel-major), w_aic [B][frameLen][M], w_bm [B*M][frameLen] — FDGSC.process's tuple */
int ds_fdgsc_process(ds_handle* h, const float* x, int n_samples, int postfilter, int dc_notch, float* out, float* p, float* fix_output,
                     float* fix_delayed, float* bm_output, float* aligned, float* aligned_delayed, float* w_aic, float* w_bm, int mem);
int ds_adaptive_frames(ds_handle* h, const float* Z, const float* gain, int n_frames, float* Y, int mem);
/* DS_ALGO_MCSPP_MVDR: x [B][n][M] (layout DS_LAYOUT_SAMPLES_CHANNELS, what the notebook hands to Transform.stft) or [B][M][n] -> y [B][n];
 * optional (NULL to skip) p [B][T][K], McSpp's speech presence probability of every frame (the notebook's `p[:, n]`) */
int ds_mcspp_mvdr_process(ds_handle* h, const float* x, int layout, int n_samples, float* y, float* p, int mem);
int ds_fdaf_update(ds_handle* h, const float* x, const float* d, const float* p, int p_mode, int n_blocks, int fir_truncate,
                   float* err, float* w_out, int mem);
int ds_omlsa_estimate(ds_handle* h, const float* y, const float* u, int n_frames, float* lambda_d, float* G, float* p, int mem);
int ds_omlsa_postfilter(ds_handle* h, const float* Y, const float* U, int n_frames, float* G, float* Yout, int mem);
int ds_sublms_update(ds_handle* h, const float* x, const float* d, const float* p, int n_frames, float* err, int mem);
int ds_subrls_update(ds_handle* h, const float* x, const float* d, int n_frames, float* err, int mem);
int ds_wpe_update(ds_handle* h, const float* x_delayed, const float* d, int n_frames, float* err, int mem);

int ds_synchronize(ds_handle* h);

/* Page-locked host memory (hipHostMalloc / hipHostFree) for buffers handed to the host-pointer entry points: transfers to and from it run at the
 * link's rate without the pages having to be faulted in and pinned per call — what a freshly allocated output array costs (10 s per call at
 * B = 1024 is 0.65 GB of output: profiles/r06a/host_api_io_groups_ab.txt).  NULL when the allocation fails. */
void* ds_host_alloc(size_t bytes);
int ds_host_free(void* p);

/* hipEvent bracket on the handle's stream (kernel timing for bench.py) */
int ds_timing_begin(ds_handle* h);
int ds_timing_end(ds_handle* h, float* elapsed_ms);

/* copy one state field to host memory; `bytes` must equal the field size */
int ds_get_state(ds_handle* h, int field, void* dst, size_t bytes);
size_t ds_field_bytes(const ds_handle* h, int field);

/* opaque checkpoint of all carried state (the reference never serialises its state) */
size_t ds_state_bytes(const ds_handle* h);
/* the carried state alone, without the checkpoint's framing and without the padding of the arrays it lives in (the lanes K .. KP - 1 of a
   plane row, the rounding of a bin's floats to float4 groups, the line padding of the RLS-WPE blocks): the bytes that carry state, which one
   call must read and write back (bench.py's byte accounting).  Smaller than ds_state_bytes by the framing and that padding */
size_t ds_state_payload_bytes(const ds_handle* h);
/* a chain handle's stage i (0-based; the order of the reference object's members, see DS_ALGO_* above): its DS_ALGO_*, channel count,
   batch and carried-state bytes (ds_state_payload_bytes of that stage).  DS_EINVAL when the handle has no stage i.  Read-only
   introspection for byte accounting (scripts/stage_budget.py) and for mapping a checkpoint to the reference object's members. */
int ds_chain_stage_info(const ds_handle* h, int i, int32_t* algo, int32_t* n_mics, int32_t* batch, size_t* payload_bytes);
/* ds_field_bytes / ds_get_state of stage i of a chain handle: the attributes the reference leaves readable on the members of a composite
   object (wpe.W / wpe.P of a Wpe: stage 1 of DS_ALGO_WPE_TD, DS_FIELD_OP_STATE; awpe.py:61-73) */
size_t ds_chain_stage_field_bytes(const ds_handle* h, int i, int field);
int ds_chain_stage_state(ds_handle* h, int i, int field, void* dst, size_t bytes);
int ds_export_state(ds_handle* h, void* dst, size_t bytes);
int ds_import_state(ds_handle* h, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* DSENH_H */
