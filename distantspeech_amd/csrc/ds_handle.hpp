// ds_handle.hpp — internal to libdsenh.so: the handle behind the C-ABI of include/dsenh.h and the helpers the three API translation
// units share (ds_api.hip: life cycle, parameters, the fused frame kernels, state; ds_api_ops.hip: frame- and block-level objects;
// ds_api_chains.hip: the chain handles DS_ALGO_WPE_MVDR / DS_ALGO_SUBBAND_GSC).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/dsenh.h"
#include "ds_kernels.hpp"
#include "ds_ops.hpp"
#include "ds_tdfilter.hpp"
#include "ds_fdaf.hpp"
#include "ds_wpe64.hpp"
#include "ds_tables.hpp"

using ds::cf;
using ds::KernelInfo;
using ds::Params;

constexpr int DS_GROUPS = 8;     // copies of a handle's device counters (one per utterance group of a chain)

struct ds_handle {
    ds_config cfg;
    int K, KP, NP, NT;
    KernelInfo ki;
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    // device state
    ds::vec4* bins;
    float* tail_in;
    float* tail_out;
    int* counters;
    ds::vec4* tables;
    cf* steer;
    int steer_per_utt;
    bool steer_set;
    // staging for host-pointer calls
    // frame-level objects (DS_ALGO_TRANSFORM .. DS_ALGO_SUBRLS)
    KernelInfo ki_istft;
    KernelInfo ki_cdr;          // DS_ALGO_SUBBAND_GSC, pipelined: the front end's analysis with McCDR as its per-bin program (null: separate kernels)
    bool front_fused;           // ... and with the DC notch and the FIR bank in front of it, ONE kernel (ds_front_kernel: shelved, make SHELVED=1 + DS_CHAIN_FRONT_FUSED=1)
    KernelInfo ki_aic;          // DS_ALGO_SUBBAND_GSC: the chain's tail as one frame kernel (null launch: separate kernels)
    KernelInfo ki_rows, ki_rows_istft;   // single-channel transform handles: the one-row-per-wavefront kernels (null launch = not available)
    int op;                     // ds::OP_* or -1
    float* opst;                // operator state [B][NF][KP]
    int NF;
    bool wpe_started;               // OP_WPE: frames went through since creation / ds_reset (or a state was imported)
    int op_frm, op_ell, op_first;   // uniform counters of the operator handle (host mirror; authoritative for host-side decisions)
    int* dev_cnt;               // device copy {frm, ell, first, aux}: aux = FIR ping-pong parity (front end) / WPE ring position (chain)
    bool use_dev_cnt;           // kernels read the counters from dev_cnt and a tick kernel follows every launch (stages of a chain handle: the
                                // launches then replay as a hipGraph)
    int filter_len, norm;
    float filt_mu, filt_alpha, rls_lambda;
    float* dev_buf[10];         // staging for host-pointer frame-level calls (3 in, 1 scratch, 5 out, 1 aux table)
    size_t dev_buf_bytes[10];
    size_t aux_floats;
    double* td_mem;             // DS_ALGO_FRONTEND: notch memories [B][M][2] doubles (the recursion runs in double: ds_ops.hpp td_dcnotch)
    float* td_cache[2];         // FIR history ping-pong [B][M][L-1]
    int td_L, td_cur;
    float* tdf_w; float* tdf_buf; float* tdf_P;     // DS_ALGO_TDNLMS / TDRLS state
    int fdaf_kind, fdaf_constrain, fdaf_non_causal, fdaf_weight_norm, fdaf_two_path;   // DS_ALGO_FDAF (state lives in opst)
    int mcspp_repeat;           // DS_PARAM_MCSPP_REPEAT
    int x_fan, p_complement, d_interleaved;   // subband LMS / RLS inside a chain: shared reference input, 1 - p, channel-interleaved d (OpParams)
    float* d_prev;              // ... and the one-frame delay line of the desired signal (owned by the chain)
    // DS_ALGO_WPE_MVDR: a chain of operator handles sharing this handle's stream, device-resident between the stages
    ds_handle* sub[10];         // WPE_MVDR: analysis transform, WPE, McMcra, adaptive frame loop, synthesis transform; SUBBAND_GSC: see chain2_*
    bool owns_stream;
    int wpe_delay;
    bool wpe_only;              // DS_ALGO_WPE_TD: analysis -> delay line -> WPE -> synthesis of channel 0 (no McMcra / MVDR stages)
    double* wpe64;              // DS_PARAM_WPE_FP64 (DS_ALGO_WPE): the recursion's state in double, [B][K][wpe64_bin_doubles] (ds_wpe64.hpp); null = fp32 kernels
    int wpe_generic;            // DS_WPE_GENERIC=1 at ds_create: every WPE shape through the run-time-shape kernels (A/B and tests)
    float* chain_buf[24];       // WPE_MVDR: D, -, E, p, G, Y, ring of the last wpe_delay analysis frames; SUBBAND_GSC: see chain2_reserve
    size_t chain_bytes[24];
    int postfilter;             // DS_PARAM_POSTFILTER (TDGSC / FDGSC chains behind ds_process_device)
    int hist_cur;               // ring slot of the oldest frame
    // cached hipGraph of a ds_process_device_seq() sequence
    hipGraphExec_t graph_exec;
    hipGraphExec_t group_exec[8];   // fused frame kernels with DS_PARAM_SPLIT > 1: one graph per free-running utterance group
    int split;                  // DS_PARAM_SPLIT: utterance groups captured as parallel graph branches
    hipStream_t side[7];        // side streams for the extra branches
    hipEvent_t ev_fork, ev_join[7];
    long long graph_key[16];
    bool graph_valid;
    // DS_ALGO_WPE_MVDR: the batch as `parts` utterance groups, each running the whole chain on its own stream (group 0: `stream`, group g:
    // side[g - 1]) at its own pace between calls — one group's WPE kernel next to the other groups' remaining stages.  Every group has
    // its own copy of the device counters (dev_cnt + 8 g); groups_open: side streams hold work the chain's stream has not joined yet
    int parts;
    bool groups_open;
    bool group_enqueue;         // inside ds_process_device_seq's per-group launches: set_device() must not join the groups it is launching
    // DS_ALGO_SUBBAND_GSC: the front end of a block (notch -> FIR bank -> analysis; latency-bound kernels) runs on its own stream
    // (side[1]) into one of two buffer sets, so that the front end of block t + 1 overlaps the HBM-bound stages of block t whenever the
    // caller has block t + 1 enqueued by then.  ev_fr: {front of set 0 / 1 done, set 0 / 1 free again}
    // ... and, with the tail as one frame kernel, the tail runs on side[2] out of one of two sets of the middle stages' outputs (p, F, the
    // blocking-matrix outputs): a three-stage pipeline front(t + 1) | McSpp + blocking filters(t) | tail(t - 1) across the blocks the caller
    // has enqueued.  ev_fr + 4: {middle of set 0 / 1 done, tail of set 0 / 1 done}; ev_fr + 8: {blocking-filter branch of set 0 / 1 done}
    // (with the tail on its own stream the branch joins there, not on the chain's stream: nothing but McSpp sits between two McSpp launches)
    bool front_async, tail_async, front_open, fr_valid[2], tf_valid[2], bf_valid[2], al_read[2], fr_mid[2];
    bool early_front;           // ... and the notch / FIR bank of block t + 2 wait for their own readers only (DS_CHAIN_NO_EARLY=1: off, A/B runs)
    bool fan_fused;             // shelved build + DS_CHAIN_FAN_FUSED=1 at ds_create: the RLS blocking filters inside McSpp's launch (OP_MCSPP_STEADY_FAN)
    bool lean_main;             // pipelined chain: McSpp's counters by value (no counter-advance launch), the blocking-filter branch joins on the tail's stream
    int front_set;
    hipEvent_t ev_fr[10];
    // chain handles under graph replay: the shape (samples per call) that has run once with plain launches (buffers sized, start-up
    // branches behind), and what one replay of the captured sequence does to the host mirrors of the stages' uniform counters
    ds_handle* owner;           // the chain handle this stage belongs to (null: stand-alone)
    ds::TickArgs pend;          // chain handle: the counter advance of the stage launched last, waiting for the next launch on pend_stream
    hipStream_t pend_stream;    // to carry it in its kernel arguments (a tick kernel of its own only if nothing follows)
    bool pend_set;
    int chain_warm_n;
    int adv_frames[11], adv_td[11], adv_hist;
    float* x_stage;
    float* y_stage;
    size_t x_stage_elems, y_stage_elems;
    // params
    int pipe_min_T;             // fused frame kernels: calls of at least this many hops run the hop-pipelined kernel (ds_pipe.hpp) where one exists
    int method;
    int mcra_L;
    float alpha_y, alpha_v, diag, gate, mu, out_scale;
    long long est_pos, est_used; // DS_PARAM_EST_POS (-1 = off) and the (frame, bin) slots consumed since the last restart (adaptivebeamformer.py:90-93)
    bool ref_powers;            // DS_PARAM_REF_POWERS (DS_ALGO_GSC): the frame kernel also writes Params::ref_pow ...
    float* ref_pow;             // ... [B][ref_pow_T][K][M] of the last call (grown on demand)
    size_t ref_pow_cap;         // floats allocated
    int ref_pow_T;              // hops of the last call (0: none yet, or the last call had no hop)
    hipStream_t ref_pow_stream; // the stream that call's kernel ran on
    std::string err;
};

namespace dsi {

extern thread_local std::string g_err;
int fail(ds_handle* h, int code, const std::string& msg);

#define DS_HIP(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return dsi::fail(h, DS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

// the fused frame kernels (one workgroup per utterance, time samples in, time samples out)
inline bool frames_algo(int algo) { return algo == DS_ALGO_FIXED || algo == DS_ALGO_ADAPTIVE || algo == DS_ALGO_GSC || algo == DS_ALGO_ADAPTIVE_PF; }
size_t bins_bytes(const ds_handle* h);
size_t tail_in_bytes(const ds_handle* h);
size_t tail_out_bytes(const ds_handle* h);
size_t opst_bytes(const ds_handle* h);
// floats between the utterances of an operator handle's state: NF rounded up to whole float4 planes (ds_ops.hpp: st_index) times KP
inline bool wpe_chain(const ds_handle* h) { return h->cfg.algo == DS_ALGO_WPE_MVDR || h->cfg.algo == DS_ALGO_WPE_TD; }
inline size_t wpe64_ust(const ds_handle* h) { return (size_t)h->K * (size_t)ds::wpe64_bin_doubles(h->cfg.n_mics, h->filter_len); }   // doubles between utterances
inline size_t wpe64_bytes(const ds_handle* h) { return h->wpe64 ? (size_t)h->cfg.batch * wpe64_ust(h) * sizeof(double) : 0; }
inline size_t op_ust(const ds_handle* h) { return (size_t)ds::st_floats_per_bin(h->NF) * h->KP; }
size_t counters_bytes(const ds_handle* h);
int set_device(ds_handle* h);
int zero_state(ds_handle* h);
void fill_params(const ds_handle* h, Params& p);
// launch the analysis / synthesis of a transform handle (Params set up for one utterance per workgroup; p.method = channels of an
// ISTFT): single-channel work goes to the one-row-per-wavefront kernels where they exist
hipError_t launch_transform_stft(const ds_handle* t, const Params& p, int batch, hipStream_t stream);
hipError_t launch_transform_istft(const ds_handle* t, const Params& p, int batch, hipStream_t stream);

// staging for host-pointer calls of the frame-level objects (ds_api_ops.hip)
struct IoSpec { const float* in[3]; size_t in_bytes[3]; float* out[5]; size_t out_bytes[5]; };
int stage_reserve(ds_handle* h, int i, size_t bytes);
int frontend_set_taps(ds_handle* h, int Lt);
int sync_dev_cnt(ds_handle* h);                 // host mirror of the uniform counters -> device copy
void advance_host_counters(ds_handle* h, int frames, int L);
// device-side counter advance behind a launch of stage t: handed to the next kernel launched on the same stream of the chain (take_tick)
int post_tick(ds_handle* t, int* cnt, int frames, int L, int aux_add, int aux_mod, hipStream_t stream);
void take_tick(ds_handle* t, hipStream_t stream, ds::TickArgs& out);
int flush_tick(ds_handle* chain);
int join_groups(ds_handle* chain);
int io_begin(ds_handle* h, int mem, const IoSpec& io, const float* din[3], float* dout[5]);
int io_end(ds_handle* h, int mem, const IoSpec& io, float* dout[5]);
int run_binop(ds_handle* h, int want_algo, const char* who, int n_frames, int mem, const IoSpec& io, int is_complex, int has_p);
// sub-range launches (utterances [b0, b0 + nb), device pointers at utterance b0) that touch no counter: the stages of a pipelined chain
int binop_launch(ds_handle* h, int b0, int nb, int n_frames, const float* const din[3], float* const dout[5], int is_complex, int has_p,
                 hipStream_t stream, const ds::TickArgs& tick, int group = 0);
int wpe_launch(ds_handle* h, int b0, int nb, const float* x_delayed, const float* d, int n_frames, float* err, float* ring, int ring_pos,
               int ring_len, const int* dev_ring_pos, hipStream_t stream, float* err0 = nullptr);
// the McSpp half of ds_mcspp_estimate on device buffers: Gamma and its band mean come from the caller (the chain's front end computes them)
// fan (optional): the chain's DS_ALGO_SUBRLS stage, run inside the same launch (OP_MCSPP_STEADY_FAN) on reference input fan_x, errors to fan_e;
// the caller advances that stage's host counters
// yout (optional): the notebook's online MVDR output of every frame, complex [B][T][K] (the full operator OP_MCSPP)
int mcspp_from_gamma(ds_handle* h, const float* y, int n_frames, const float* gamma, const float* qavg, float* p_out, ds_handle* fan = nullptr,
                     const float* fan_x = nullptr, float* fan_e = nullptr, float* yout = nullptr);
int wpe_run(ds_handle* h, const float* x_delayed, const float* d, int n_frames, float* err, int mem, float* ring, int ring_pos, int ring_len,
            const int* dev_ring_pos, float* err0 = nullptr);

// chain handles (ds_api_chains.hip)
// DS_ALGO_SUBBAND_GSC: device buffers of the chain (indices into ds_handle::chain_buf; "c" = complex64)
enum {
    G_XN = 0,        // [B][M][n]        input after the DC notch
    G_XA = 1,        // [B][M][n]        time-aligned channels
    G_FIXED = 2,     // [B][n]           fixed beamformer output (channel mean)
    G_D = 3,         // c[B][T][K][M]    STFT of the aligned channels
    G_P = 4,         // [B][T][K]        McSpp speech presence probability
    G_F = 6,         // c[B][T][K]       STFT of the fixed beamformer output
    G_E = 8,         // c[B*M][T][K]     blocking-filter errors
    G_BM = 9,        // [B][M][n]        blocking-matrix outputs in the time domain
    G_XAIC = 10,     // c[B][T][K][M]    their STFT: the canceller's input
    G_E2 = 12,       // c[B][T][K]       canceller error = output spectrum
    G_FPREV = 13,    // c[B][K]          state: F of the previous block (delay_fbf in the spectral domain)
    G_FIXPREV = 14,  // [B][hop]         state: fixed beamformer output of the previous block
    G_XN2 = 15, G_XA2 = 16, G_FIXED2 = 17, G_D2 = 18,   // second set of the front end's buffers (front_async)
    G_P2 = 19, G_F2 = 20, G_EB2 = 21,                   // ... and of what the tail reads (p, F, the blocking filters' error spectra)
    G_GAM = 22, G_GAM2 = 23,                            // McCDR's Gamma [B][T][K] + its band mean [B][T] when the front end computes them (ki_cdr), two sets
    G_COUNT = 24
};
int chain_reserve(ds_handle* h, int T);
int chain_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                         int n_samples, float* y_dev, long long y_batch_stride);
int chain2_reserve(ds_handle* h, int n);
// DS_ALGO_MCSPP_MVDR (ds_api_chains.hip): analysis -> McSpp + steering + MVDR -> synthesis on device buffers; p_dev optional [B][T][K]
int nbmvdr_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride, int n_samples,
                          float* y_dev, long long y_batch_stride, float* p_dev);
// DS_ALGO_TDGSC / DS_ALGO_FDGSC (ds_api_gsc_chains.hip)
int gsc_chain_create(ds_handle* h);
int tdgsc_run(ds_handle* h, const float* x, long long x_bstride, long long x_cstride, int n, int postfilter, float* out, long long out_bstride,
              float* p_dev, float* bm_dev, float* w_dev);
int fdgsc_run(ds_handle* h, const float* x, long long x_bstride, long long x_cstride, int n, int postfilter, int dc_notch, float* out,
              long long out_bstride, float* p_dev, float* fix_dev, float* fixd_dev, float* bm_dev, float* al_dev, float* ald_dev,
              float* waic_dev, float* wbm_dev);
// overlap-save FDAF launch on device pointers with the chain addressing options (x_fan / strides; 0 = dense)
int fdaf_run_dev(ds_handle* h, const float* x, const float* d, const float* pp, int p_mode, int n_blocks, int fir_truncate, float* err,
                 float* w_out, int x_fan, long long x_inst_stride, long long x_sample_stride, long long x_chan_stride);
int chain2_run(ds_handle* h, const float* x_dev, long long x_bstride, long long x_cstride, int n, float* y_dev, long long y_bstride,
               float* fix_dev, float* bm_dev, float* p_dev, float* al_dev);

}  // namespace dsi
