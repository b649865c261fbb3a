// ds_api_ops.hip — C-ABI entry points of the frame- and block-level objects (include/dsenh.h, "frame-level entry points"):
// staging of host-pointer calls, the per-(utterance, bin) operator launches, the block kernels (FDAF, WPE, time-domain filters).
#include "ds_handle.hpp"

using namespace dsi;

namespace dsi {

// make sure staging slot `i` holds at least `bytes`
// FIR history of a front-end handle for an L-tap bank: two ping-pong buffers [B][L-1][M], zeroed; no-op when L is unchanged
int frontend_set_taps(ds_handle* h, int Lt) {
    if (h->td_L == Lt) return DS_OK;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    const size_t cb = (size_t)h->cfg.batch * (Lt > 1 ? Lt - 1 : 1) * h->cfg.n_mics * sizeof(float);
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(h->td_cache[i]); h->td_cache[i] = nullptr;
        DS_HIP(h, hipMalloc((void**)&h->td_cache[i], cb));
        DS_HIP(h, hipMemset(h->td_cache[i], 0, cb));
    }
    h->td_L = Lt; h->td_cur = 0;
    return sync_dev_cnt(h);
}

int fdaf_run_dev(ds_handle* h, const float* x, const float* d, const float* pp, int p_mode, int n_blocks, int fir_truncate, float* err,
                 float* w_out, int x_fan, long long x_inst_stride, long long x_sample_stride, long long x_chan_stride) {
    ds::FdafParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.T = n_blocks; p.C = h->cfg.n_mics;
    p.kind = h->fdaf_kind; p.constrain = h->fdaf_constrain; p.non_causal = h->fdaf_non_causal; p.weight_norm = h->fdaf_weight_norm;
    p.two_path = h->fdaf_two_path && h->fdaf_kind == DS_FDAF_PLAIN;
    p.trunc = fir_truncate < 0 ? -1 : fir_truncate; p.p_mode = p_mode & 3; p.p_complement = (p_mode & DS_FDAF_P_COMPLEMENT) ? 1 : 0;
    p.mu = h->filt_mu; p.alpha = h->filt_alpha;
    p.x = x; p.d = d; p.p = pp; p.err = err; p.w_out = w_out;
    p.x_fan = x_fan; p.x_inst_stride = x_inst_stride; p.x_sample_stride = x_sample_stride; p.x_chan_stride = x_chan_stride;
    p.state = h->opst; p.state_stride = (long long)op_ust(h);
    p.tables = h->tables;
    DS_HIP(h, ds::launch_fdaf(p, h->cfg.nfft, h->stream));
    return DS_OK;
}

int stage_reserve(ds_handle* h, int i, size_t bytes) {
    if (bytes <= h->dev_buf_bytes[i]) return DS_OK;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->dev_buf[i]); h->dev_buf[i] = nullptr; h->dev_buf_bytes[i] = 0;
    DS_HIP(h, hipMalloc((void**)&h->dev_buf[i], bytes));
    h->dev_buf_bytes[i] = bytes;
    return DS_OK;
}


// resolve host/device pointers: for DS_MEM_HOST copy inputs to staging and return device aliases
int io_begin(ds_handle* h, int mem, const IoSpec& io, const float* din[3], float* dout[5]) {
    for (int i = 0; i < 5; ++i) {
        if (i < 3) din[i] = io.in[i];
        dout[i] = io.out[i];
        if (mem == DS_MEM_HOST) {
            if (i < 3 && io.in[i]) {
                int rc = stage_reserve(h, i, io.in_bytes[i]); if (rc) return rc;
                DS_HIP(h, hipMemcpyAsync(h->dev_buf[i], io.in[i], io.in_bytes[i], hipMemcpyHostToDevice, h->stream));
                din[i] = h->dev_buf[i];
            }
            if (io.out[i]) {
                int rc = stage_reserve(h, 4 + i, io.out_bytes[i]); if (rc) return rc;
                dout[i] = h->dev_buf[4 + i];
            }
        }
    }
    return DS_OK;
}

int io_end(ds_handle* h, int mem, const IoSpec& io, float* dout[5]) {
    if (mem == DS_MEM_HOST) {
        for (int i = 0; i < 5; ++i)
            if (io.out[i]) DS_HIP(h, hipMemcpyAsync(io.out[i], dout[i], io.out_bytes[i], hipMemcpyDeviceToHost, h->stream));
        DS_HIP(h, hipStreamSynchronize(h->stream));
    }
    return DS_OK;
}

// one launch of the operator over utterances [b0, b0 + nb) of the handle's batch; the io pointers already point at utterance b0.
// Touches no counter: the caller advances the host mirrors and posts the device tick once the whole batch has been launched.
int binop_launch(ds_handle* h, int b0, int nb, int n_frames, const float* const din[3], float* const dout[5], int is_complex, int has_p,
                 hipStream_t stream, const ds::TickArgs& tick, int group) {
    if (b0 != 0 && h->d_prev) return fail(h, DS_EUNSUPPORTED, "binop_launch: utterance sub-ranges are not available with a delayed desired signal");
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = nb; p.K = h->K; p.KP = h->KP; p.T = n_frames;
    p.st = h->opst + (size_t)b0 * op_ust(h); p.NF = h->NF;
    p.in0 = din[0]; p.in1 = din[1]; p.in2 = din[2];
    p.out0 = dout[0]; p.out1 = dout[1]; p.out2 = dout[2]; p.out3 = dout[3]; p.out4 = dout[4];
    p.M = h->cfg.n_mics; p.N = h->filter_len;
    p.frm_cnt = h->op_frm; p.ell = h->op_ell; p.L = h->mcra_L; p.first_frame = h->op_first;
    p.in_complex = is_complex; p.has_p = has_p; p.norm = h->norm;
    p.mu = h->filt_mu; p.alpha = h->filt_alpha; p.reg = 1e-4f; p.lam = h->rls_lambda;
    p.x_fan = h->x_fan > 0 ? h->x_fan : 1; p.p_complement = h->p_complement; p.d_interleaved = h->d_interleaved; p.d_prev = h->d_prev;
    p.steer_batch_stride = h->steer_per_utt ? (long long)h->K * h->cfg.n_mics : 0;
    p.steer = h->steer ? h->steer + (size_t)b0 * p.steer_batch_stride : nullptr;
    p.method = h->method; p.alpha_v = h->alpha_v; p.beta_v = ds::complement_of(h->alpha_v); p.gate = h->gate; p.diag = h->diag; p.diag_floor = ds::pivot_floor(h->diag);
    p.dev_cnt = h->use_dev_cnt ? h->dev_cnt + 8 * group : nullptr;
    p.tick = tick;
    DS_HIP(h, ds::launch_binop(h->op, p, stream));
    return DS_OK;
}

int run_binop(ds_handle* h, int want_algo, const char* who, int n_frames, int mem, const IoSpec& io, int is_complex, int has_p) {
    if (!h) return DS_EINVAL;
    if (h->cfg.algo != want_algo) return fail(h, DS_ESTATE, std::string(who) + ": handle was created for a different algo");
    if (n_frames < 0) return fail(h, DS_ESHAPE, std::string(who) + ": n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::TickArgs tick;
    take_tick(h, h->stream, tick);                                     // an earlier stage's counter advance rides in this launch
    rc = binop_launch(h, 0, h->cfg.batch, n_frames, din, dout, is_complex, has_p, h->stream, tick); if (rc) return rc;
    // advance the uniform counters exactly like the kernel did (mcra.py:52-56,72-74): the host mirror, and the device copy behind the launch
    // (only the operators that read them: the subband filters keep no frame counters)
    const bool counts = h->op != ds::OP_SUBLMS && h->op != ds::OP_SUBRLS;
    if (h->use_dev_cnt && counts) { rc = post_tick(h, h->dev_cnt, n_frames, h->mcra_L, 0, 0, h->stream); if (rc) return rc; }
    advance_host_counters(h, n_frames, h->mcra_L);
    return io_end(h, mem, io, dout);
}

// the WPE kernel over utterances [b0, b0 + nb) (device pointers at utterance b0; ring = the chain's delay line at utterance b0 or null)
int wpe_launch(ds_handle* h, int b0, int nb, const float* x_delayed, const float* d, int n_frames, float* err, float* ring, int ring_pos,
               int ring_len, const int* dev_ring_pos, hipStream_t stream, float* err0) {
    ds::WpeParams p;
    std::memset(&p, 0, sizeof p);
    p.B = nb; p.K = h->K; p.T = n_frames; p.C = h->cfg.n_mics; p.N = h->filter_len;
    p.xd = x_delayed; p.d = d; p.err = err; p.lam = h->rls_lambda;
    p.ustride = (long long)op_ust(h);
    p.state = h->opst + (size_t)b0 * p.ustride;
    p.ring = ring; p.ring_pos = ring_pos; p.ring_len = ring_len; p.dev_ring_pos = dev_ring_pos; p.err0 = err0;
    h->wpe_started = true;                                  // DS_PARAM_WPE_FP64 is refused from here until ds_reset
    if (h->wpe64) {                                        // DS_PARAM_WPE_FP64: the whole recursion in double (ds_wpe64.hpp)
        ds::Wpe64Params q;
        q.w = p;
        q.ustride64 = (long long)wpe64_ust(h); q.lam64 = ds::wpe64_lambda(h->rls_lambda);
        q.state64 = h->wpe64 + (size_t)b0 * q.ustride64;
        DS_HIP(h, ds::launch_wpe64(q, stream));
        return DS_OK;
    }
    if (p.C * p.N > ds::WPE_CNMAX) DS_HIP(h, ds::launch_wpe_wide(p, h->wpe_generic, stream));     // wide prediction filters: one wavefront per bin
    else DS_HIP(h, ds::launch_wpe(p, h->wpe_generic, stream));
    return DS_OK;
}

int wpe_run(ds_handle* h, const float* x_delayed, const float* d, int n_frames, float* err, int mem, float* ring, int ring_pos, int ring_len,
            const int* dev_ring_pos, float* err0) {
    if (!h || (!x_delayed && !ring) || !d || !err) return fail(h, DS_EINVAL, "ds_wpe_update: NULL argument");
    if (h->cfg.algo != DS_ALGO_WPE) return fail(h, DS_ESTATE, "ds_wpe_update: handle was created for a different algo");
    if (n_frames < 0) return fail(h, DS_ESHAPE, "ds_wpe_update: n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_frames * h->K * h->cfg.n_mics * 8;
    IoSpec io = {{x_delayed, d, nullptr}, {x_delayed ? n : 0, n, 0}, {err, nullptr, nullptr}, {n, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    rc = wpe_launch(h, 0, h->cfg.batch, din[0], din[1], n_frames, dout[0], ring, ring_pos, ring_len, dev_ring_pos, h->stream, err0); if (rc) return rc;
    return io_end(h, mem, io, dout);
}

}  // namespace dsi

extern "C" {

int ds_stft(ds_handle* h, const float* x, int layout, int n_samples, float* Y, int mem) {
    if (!h || !x || !Y) return fail(h, DS_EINVAL, "ds_stft: NULL argument");
    if (h->cfg.algo != DS_ALGO_TRANSFORM) return fail(h, DS_ESTATE, "ds_stft: handle is not a DS_ALGO_TRANSFORM object");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_stft: n_samples must be a multiple of hop");
    if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES) return fail(h, DS_EINVAL, "ds_stft: unknown layout");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, T = n_samples / h->cfg.hop;
    IoSpec io = {{x, nullptr, nullptr}, {B * M * (size_t)n_samples * 4, 0, 0}, {Y, nullptr, nullptr}, {B * T * h->K * M * 8, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    Params p;
    fill_params(h, p);
    p.x = din[0]; p.y = dout[0];
    p.x_batch_stride = (long long)(M * (size_t)n_samples);
    p.y_batch_stride = (long long)(T * h->K * M * 2);
    if (layout == DS_LAYOUT_CHANNELS_SAMPLES) { p.x_sample_stride = 1; p.x_chan_stride = n_samples; }
    else { p.x_sample_stride = (long long)M; p.x_chan_stride = 1; }
    p.T = (int)T; p.batch0 = 0;
    DS_HIP(h, launch_transform_stft(h, p, h->cfg.batch, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_istft(ds_handle* h, const float* Y, int n_frames, int n_channels, float* y, int mem) {
    if (!h || !Y || !y) return fail(h, DS_EINVAL, "ds_istft: NULL argument");
    if (h->cfg.algo != DS_ALGO_TRANSFORM) return fail(h, DS_ESTATE, "ds_istft: handle is not a DS_ALGO_TRANSFORM object");
    if (n_channels < 1 || n_channels > h->cfg.n_mics)                       // transform.py:466
        return fail(h, DS_ESHAPE, "ds_istft: n_channels must be in 1..channel");
    if (n_frames < 0) return fail(h, DS_ESHAPE, "ds_istft: n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, C = n_channels, T = n_frames;
    IoSpec io = {{Y, nullptr, nullptr}, {B * T * h->K * C * 8, 0, 0}, {y, nullptr, nullptr}, {B * T * h->cfg.hop * C * 4, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    Params p;
    fill_params(h, p);
    p.x = din[0]; p.y = dout[0];
    p.x_batch_stride = (long long)(T * h->K * C * 2);
    p.y_batch_stride = (long long)(T * h->cfg.hop * C);
    p.T = (int)T; p.batch0 = 0; p.method = n_channels;
    DS_HIP(h, launch_transform_istft(h, p, h->cfg.batch, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_mcra_estimate_p(ds_handle* h, const float* Y, int is_complex, int n_frames, float* lambda_d, float* p, int mem) {
    if (!h || !Y || !lambda_d) return fail(h, DS_EINVAL, "ds_mcra_estimate: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{Y, nullptr, nullptr}, {n * (is_complex ? 8 : 4), 0, 0}, {lambda_d, p, nullptr}, {n * 4, p ? n * 4 : 0, 0}};
    return run_binop(h, DS_ALGO_MCRA, "ds_mcra_estimate", n_frames, mem, io, is_complex ? 1 : 0, 0);
}

int ds_mcra_estimate(ds_handle* h, const float* Y, int is_complex, int n_frames, float* lambda_d, int mem) {
    return ds_mcra_estimate_p(h, Y, is_complex, n_frames, lambda_d, nullptr, mem);
}

int ds_mcmcra_estimate(ds_handle* h, const float* y, int n_frames, float* p, float* G, int mem) {
    if (!h || !y || !p || !G) return fail(h, DS_EINVAL, "ds_mcmcra_estimate: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{y, nullptr, nullptr}, {n * h->cfg.n_mics * 8, 0, 0}, {p, G, nullptr}, {n * 4, n * 4, 0}};
    return run_binop(h, DS_ALGO_MCMCRA, "ds_mcmcra_estimate", n_frames, mem, io, 0, 0);
}

int ds_mcsppbase_estimate(ds_handle* h, const float* y, int n_frames, float* p, float* w, int mem) {
    if (!h || !y || !p || !w) return fail(h, DS_EINVAL, "ds_mcsppbase_estimate: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{y, nullptr, nullptr}, {n * h->cfg.n_mics * 8, 0, 0}, {p, w, nullptr}, {n * 4, n * h->cfg.n_mics * 8, 0}};
    return run_binop(h, DS_ALGO_MCSPPBASE, "ds_mcsppbase_estimate", n_frames, mem, io, 0, 0);
}

int ds_set_aux(ds_handle* h, const float* table, size_t n_floats) {
    if (!h || !table || n_floats == 0) return fail(h, DS_EINVAL, "ds_set_aux: NULL argument");
    int rc = set_device(h); if (rc) return rc;
    rc = stage_reserve(h, 9, n_floats * sizeof(float)); if (rc) return rc;
    DS_HIP(h, hipMemcpy(h->dev_buf[9], table, n_floats * sizeof(float), hipMemcpyHostToDevice));
    h->aux_floats = n_floats;
    // a front-end handle's table is its FIR bank coef[L][M]: the history (and with it the size of the exported state) is fixed here, not
    // at the first filtering call, so a blob exported from a running handle imports into a freshly configured one
    if (h->cfg.algo == DS_ALGO_FRONTEND && n_floats % (size_t)h->cfg.n_mics == 0) {
        rc = frontend_set_taps(h, (int)(n_floats / h->cfg.n_mics)); if (rc) return rc;
    }
    return DS_OK;
}

int ds_mcspp_estimate(ds_handle* h, const float* y, int n_frames, float* p_out, float* w_pmwf, float* yout, float* phi_xx,
                      float* phi_vv_inv, int mem) {
    if (!h || !y || !p_out) return fail(h, DS_EINVAL, "ds_mcspp_estimate: NULL argument");
    if (h->cfg.algo != DS_ALGO_MCSPP) return fail(h, DS_ESTATE, "ds_mcspp_estimate: handle was created for a different algo");
    if ((phi_xx == nullptr) != (phi_vv_inv == nullptr)) return fail(h, DS_EINVAL, "ds_mcspp_estimate: phi_xx and phi_vv_inv go together");
    if (h->aux_floats < (size_t)h->K) return fail(h, DS_ESTATE, "ds_mcspp_estimate: call ds_set_aux(h, Fn[K]) first");
    if (n_frames < 0) return fail(h, DS_ESHAPE, "ds_mcspp_estimate: n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_frames * h->K, M = h->cfg.n_mics;
    IoSpec io = {{y, nullptr, nullptr}, {n * M * 8, 0, 0}, {p_out, w_pmwf, yout, phi_xx, phi_vv_inv},
                 {n * 4, w_pmwf ? n * M * 8 : 0, yout ? n * 8 : 0, phi_xx ? n * M * M * 8 : 0, phi_vv_inv ? n * M * M * 8 : 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    const size_t nbt = (size_t)h->cfg.batch * n_frames;
    rc = stage_reserve(h, 3, (n + nbt) * 4); if (rc) return rc;              // Gamma [B][T][K], then its band mean [B][T]
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.KP = h->KP; p.T = n_frames; p.st = h->opst; p.NF = h->NF; p.M = h->cfg.n_mics;
    p.frm_cnt = h->op_frm; p.ell = h->op_ell; p.L = 65;                      // mccdr.py:60-61
    p.dev_cnt = h->use_dev_cnt ? h->dev_cnt : nullptr;
    p.in0 = din[0]; p.in1 = h->dev_buf[9]; p.out0 = h->dev_buf[3];
    take_tick(h, h->stream, p.tick);
    DS_HIP(h, ds::launch_binop(ds::OP_MCCDR, p, h->stream));
    p.tick = ds::TickArgs{nullptr, 0, 1, 0, 0};
    // the band average of the prior, one value per (utterance, frame): formed inside the McSpp kernel's own prologue for short calls
    // (ds_binop_kernel), by a launch of its own otherwise
    const int band_n = (int)(2000.0 * (2 * (h->K - 1)) / 16000.0) - (int)(500.0 * (2 * (h->K - 1)) / 16000.0);
    const bool qavg_in_kernel = n_frames <= 64 && band_n > 0 && band_n <= 64;
    if (!qavg_in_kernel) DS_HIP(h, ds::launch_mcspp_qavg(h->dev_buf[3], h->dev_buf[3] + n, (int)nbt, h->K, h->stream));
    p.in1 = h->dev_buf[3]; p.in2 = qavg_in_kernel ? nullptr : h->dev_buf[3] + n;      // (the McSpp rows start at the constexpr MCSPP_ROW0: nothing to pass)
    p.out0 = dout[0]; p.out1 = dout[1]; p.out2 = dout[2]; p.out3 = dout[3]; p.out4 = dout[4];
    // three builds of the same estimation: with the notebook-MVDR / matrix outputs, lean (p and the optional PMWF weights), and the lean
    // one for calls that start at frame 5 or later without weights (no second factorisation in the kernel: two waves per SIMD at 6 mics)
    p.repeat = h->mcspp_repeat;
    const int op = (yout || phi_xx || p.repeat) ? ds::OP_MCSPP : (w_pmwf || h->op_frm < 5) ? ds::OP_MCSPP_LEAN : ds::OP_MCSPP_STEADY;
    DS_HIP(h, ds::launch_binop(op, p, h->stream));
    if (h->use_dev_cnt) { rc = post_tick(h, h->dev_cnt, n_frames, 65, 0, 0, h->stream); if (rc) return rc; }
    { const int first = h->op_first; advance_host_counters(h, n_frames, 65); h->op_first = first; }
    return io_end(h, mem, io, dout);
}

}  // extern "C"
namespace dsi {
int mcspp_from_gamma(ds_handle* h, const float* y, int n_frames, const float* gamma, const float* qavg, float* p_out, ds_handle* fan,
                     const float* fan_x, float* fan_e, float* yout) {
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.KP = h->KP; p.T = n_frames; p.st = h->opst; p.NF = h->NF; p.M = h->cfg.n_mics;
    p.frm_cnt = h->op_frm; p.ell = h->op_ell; p.L = 65;
    p.dev_cnt = h->use_dev_cnt ? h->dev_cnt : nullptr;
    p.in0 = y; p.in1 = gamma; p.in2 = qavg;
    p.out0 = p_out; p.out2 = yout;
    p.repeat = h->mcspp_repeat;
    take_tick(h, h->stream, p.tick);
    int op = (p.repeat || yout) ? ds::OP_MCSPP : h->op_frm < 5 ? ds::OP_MCSPP_LEAN : ds::OP_MCSPP_STEADY;
    if (fan) {                                           // the chain's RLS blocking filters in the same threads (mcspp_fan_ok() said so)
        if (op != ds::OP_MCSPP_STEADY) return fail(h, DS_ESTATE, "mcspp_from_gamma: the fused blocking filters need the steady-state McSpp build");
        op = ds::OP_MCSPP_STEADY_FAN;
        p.fan_st = fan->opst; p.fan_NF = fan->NF; p.fan_x = fan_x; p.fan_e = fan_e; p.fan_lam = fan->rls_lambda; p.fan_mu = fan->filt_mu;
    }
    DS_HIP(h, ds::launch_binop(op, p, h->stream));
    if (h->use_dev_cnt) { const int rc = post_tick(h, h->dev_cnt, n_frames, 65, 0, 0, h->stream); if (rc) return rc; }
    { const int first = h->op_first; advance_host_counters(h, n_frames, 65); h->op_first = first; }
    return DS_OK;
}
}  // namespace dsi
extern "C" {

static int run_linalg(ds_handle* h, int op, const char* who, const IoSpec& io, int mem, float mu = 0.0f, float reg = 0.0f) {
    if (h->cfg.algo != DS_ALGO_LINALG) return fail(h, DS_ESTATE, std::string(who) + ": handle is not a DS_ALGO_LINALG object");
    int rc = set_device(h); if (rc) return rc;
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.KP = h->KP; p.T = 1; p.M = h->cfg.n_mics;
    p.in0 = din[0]; p.in1 = din[1]; p.in2 = din[2]; p.out0 = dout[0];
    p.mu = mu; p.reg = reg;
    DS_HIP(h, ds::launch_binop(op, p, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_steering(ds_handle* h, const float* XX, float* v, int mem) {
    if (!h || !XX || !v) return fail(h, DS_EINVAL, "ds_steering: NULL argument");
    const size_t n = (size_t)h->cfg.batch * h->K, M = h->cfg.n_mics;
    IoSpec io = {{XX, nullptr, nullptr}, {n * M * M * 8, 0, 0}, {v, nullptr, nullptr, nullptr, nullptr}, {n * M * 8, 0, 0, 0, 0}};
    return run_linalg(h, ds::OP_STEERING, "ds_steering", io, mem);
}

int ds_mvdr_weight(ds_handle* h, const float* steer, const float* Rinv, float* w, int mem) {
    if (!h || !steer || !Rinv || !w) return fail(h, DS_EINVAL, "ds_mvdr_weight: NULL argument");
    const size_t n = (size_t)h->cfg.batch * h->K, M = h->cfg.n_mics;
    IoSpec io = {{steer, Rinv, nullptr}, {n * M * 8, n * M * M * 8, 0}, {w, nullptr, nullptr, nullptr, nullptr}, {n * M * 8, 0, 0, 0, 0}};
    return run_linalg(h, ds::OP_MVDRW, "ds_mvdr_weight", io, mem);
}

int ds_pmwf_weight(ds_handle* h, const float* xi, const float* Rxx, const float* Rvv_inv, float beta, float* w, int mem) {
    if (!h || !xi || !Rxx || !Rvv_inv || !w) return fail(h, DS_EINVAL, "ds_pmwf_weight: NULL argument");
    const size_t n = (size_t)h->cfg.batch * h->K, M = h->cfg.n_mics;
    IoSpec io = {{xi, Rxx, Rvv_inv}, {n * 4, n * M * M * 8, n * M * M * 8}, {w, nullptr, nullptr, nullptr, nullptr}, {n * M * 8, 0, 0, 0, 0}};
    return run_linalg(h, ds::OP_PMWFW, "ds_pmwf_weight", io, mem, beta, 0.0f);
}

int ds_gev_vector(ds_handle* h, const float* target, const float* noise, float* v, int mem) {
    if (!h || !target || !noise || !v) return fail(h, DS_EINVAL, "ds_gev_vector: NULL argument");
    const size_t n = (size_t)h->cfg.batch * h->K, M = h->cfg.n_mics;
    IoSpec io = {{target, noise, nullptr}, {n * M * M * 8, n * M * M * 8, 0}, {v, nullptr, nullptr, nullptr, nullptr}, {n * M * 8, 0, 0, 0, 0}};
    return run_linalg(h, ds::OP_GEV, "ds_gev_vector", io, mem);
}

int ds_blind_analytic_normalization(ds_handle* h, const float* vector, const float* noise, float eps, float* out, int mem) {
    if (!h || !vector || !noise || !out) return fail(h, DS_EINVAL, "ds_blind_analytic_normalization: NULL argument");
    const size_t n = (size_t)h->cfg.batch * h->K, M = h->cfg.n_mics;
    IoSpec io = {{vector, noise, nullptr}, {n * M * 8, n * M * M * 8, 0}, {out, nullptr, nullptr, nullptr, nullptr}, {n * M * 8, 0, 0, 0, 0}};
    return run_linalg(h, ds::OP_BAN, "ds_blind_analytic_normalization", io, mem, 0.0f, eps);
}

int ds_phase_correction(ds_handle* h, const float* vector, float* out, int mem) {
    if (!h || !vector || !out) return fail(h, DS_EINVAL, "ds_phase_correction: NULL argument");
    const size_t n = (size_t)h->cfg.batch * h->K, M = h->cfg.n_mics;
    IoSpec io = {{vector, nullptr, nullptr}, {n * M * 8, 0, 0}, {out, nullptr, nullptr, nullptr, nullptr}, {n * M * 8, 0, 0, 0, 0}};
    return run_linalg(h, ds::OP_PHASECORR, "ds_phase_correction", io, mem);
}

int ds_dcnotch(ds_handle* h, const float* x, int n_samples, float* y, int mem) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_dcnotch: NULL argument");
    if (h->cfg.algo != DS_ALGO_FRONTEND) return fail(h, DS_ESTATE, "ds_dcnotch: handle is not a DS_ALGO_FRONTEND object");
    if (n_samples < 0) return fail(h, DS_ESHAPE, "ds_dcnotch: n_samples < 0");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * h->cfg.n_mics * n_samples;
    IoSpec io = {{x, nullptr, nullptr}, {n * 4, 0, 0}, {y, nullptr, nullptr, nullptr, nullptr}, {n * 4, 0, 0, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::TdParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.M = h->cfg.n_mics; p.n = n_samples; p.x = din[0]; p.y = dout[0]; p.mem = h->td_mem;
    p.radius = ds::decimal_double(h->cfg.filt_alpha > 0 ? h->cfg.filt_alpha : 0.9f);
    DS_HIP(h, ds::launch_dcnotch(p, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_firbank_bm(ds_handle* h, const float* x, int n_samples, float* y, float* mean, float* bm, int mem) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_firbank_bm: NULL argument");
    if (h->cfg.algo != DS_ALGO_FRONTEND) return fail(h, DS_ESTATE, "ds_firbank_bm: handle is not a DS_ALGO_FRONTEND object");
    const int M = h->cfg.n_mics;
    if (h->aux_floats == 0 || h->aux_floats % M != 0) return fail(h, DS_ESTATE, "ds_firbank_bm: call ds_set_aux(h, coef[L][M]) first");
    const int Lt = (int)(h->aux_floats / M);
    if (n_samples < 0) return fail(h, DS_ESHAPE, "ds_firbank_bm: n_samples < 0");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    rc = frontend_set_taps(h, Lt); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_samples;
    IoSpec io = {{x, nullptr, nullptr}, {n * M * 4, 0, 0}, {y, mean, bm, nullptr, nullptr}, {n * M * 4, mean ? n * 4 : 0, bm ? n * (M - 1) * 4 : 0, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::TdParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.M = M; p.n = n_samples; p.L = Lt; p.x = din[0]; p.y = dout[0]; p.mean = dout[1]; p.diff = bm ? dout[2] : nullptr;
    p.coef = h->dev_buf[9]; p.cache_in = h->td_cache[h->td_cur]; p.cache_out = h->td_cache[h->td_cur ^ 1];
    DS_HIP(h, ds::launch_fir(p, h->stream));
    h->td_cur ^= 1;
    return io_end(h, mem, io, dout);
}

int ds_firbank(ds_handle* h, const float* x, int n_samples, float* y, float* mean, int mem) {
    return ds_firbank_bm(h, x, n_samples, y, mean, nullptr, mem);
}

int ds_tdfilter_update(ds_handle* h, const float* x, const float* d, int n_samples, float p_upd, float* err, int mem) {
    if (!h || !x || !d || !err) return fail(h, DS_EINVAL, "ds_tdfilter_update: NULL argument");
    if (h->cfg.algo != DS_ALGO_TDNLMS && h->cfg.algo != DS_ALGO_TDRLS)
        return fail(h, DS_ESTATE, "ds_tdfilter_update: handle is not a DS_ALGO_TDNLMS / DS_ALGO_TDRLS object");
    if (n_samples < 0) return fail(h, DS_ESHAPE, "ds_tdfilter_update: n_samples < 0");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_samples;
    IoSpec io = {{x, d, nullptr}, {n * 4, n * 4, 0}, {err, nullptr, nullptr, nullptr, nullptr}, {n * 4, 0, 0, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::TdfParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.n = n_samples; p.L = h->cfg.filter_len;
    p.mode = h->cfg.algo == DS_ALGO_TDRLS ? ds::TDF_RLS : ds::TDF_NLMS;
    p.x = din[0]; p.d = din[1]; p.err = dout[0]; p.w = h->tdf_w; p.buf = h->tdf_buf; p.P = h->tdf_P;
    p.mu = h->filt_mu; p.eps = 1e-4f; p.p = p_upd; p.lam = h->rls_lambda; p.norm = h->norm;
    DS_HIP(h, ds::launch_tdfilter(p, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_adaptive_frames(ds_handle* h, const float* Z, const float* gain, int n_frames, float* Y, int mem) {
    if (!h || !Z || !Y) return fail(h, DS_EINVAL, "ds_adaptive_frames: NULL argument");
    if (h->cfg.algo == DS_ALGO_ADAPTIVE_FRAMES && !h->steer_set) return fail(h, DS_ESTATE, "ds_adaptive_frames: call ds_set_steering first");
    if (h->method == DS_METHOD_TFGSC) return fail(h, DS_EUNSUPPORTED, "ds_adaptive_frames: TFGSC needs Ryy, use the fused DS_ALGO_ADAPTIVE kernel");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{Z, gain, nullptr}, {n * h->cfg.n_mics * 8, gain ? n * 4 : 0, 0}, {Y, nullptr, nullptr}, {n * 8, 0, 0}};
    return run_binop(h, DS_ALGO_ADAPTIVE_FRAMES, "ds_adaptive_frames", n_frames, mem, io, 0, gain ? 1 : 0);
}

int ds_fdaf_update(ds_handle* h, const float* x, const float* d, const float* pp, int p_mode, int n_blocks, int fir_truncate,
                   float* err, float* w_out, int mem) {
    if (!h || !x || !d || !err) return fail(h, DS_EINVAL, "ds_fdaf_update: NULL argument");
    if (h->cfg.algo != DS_ALGO_FDAF) return fail(h, DS_ESTATE, "ds_fdaf_update: handle is not a DS_ALGO_FDAF object");
    if (n_blocks < 0) return fail(h, DS_ESHAPE, "ds_fdaf_update: n_blocks < 0");
    const int p_kind = p_mode & 3;
    if (p_mode < 0 || (p_mode & ~(3 | DS_FDAF_P_COMPLEMENT)) || p_kind > DS_FDAF_P_BIN || (p_kind != DS_FDAF_P_NONE && !pp))
        return fail(h, DS_EINVAL, "ds_fdaf_update: p_mode / p mismatch");
    const int L = h->cfg.nfft / 2, C = h->cfg.n_mics;
    if (fir_truncate > L) return fail(h, DS_ESHAPE, "ds_fdaf_update: fir_truncate > filter_len");
    if (n_blocks == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_blocks * L;
    const size_t pbytes = p_kind == DS_FDAF_P_NONE ? 0 : (size_t)h->cfg.batch * n_blocks * (p_kind == DS_FDAF_P_BIN ? h->K : 1) * 4;
    IoSpec io = {{x, d, p_kind == DS_FDAF_P_NONE ? nullptr : pp}, {n * C * 4, n * 4, pbytes},
                 {err, w_out, nullptr, nullptr, nullptr}, {n * 4, w_out ? (size_t)h->cfg.batch * L * C * 4 : 0, 0, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    rc = fdaf_run_dev(h, din[0], din[1], din[2], p_mode, n_blocks, fir_truncate, dout[0], w_out ? dout[1] : nullptr, 0, 0, 0, 0);
    if (rc) return rc;
    return io_end(h, mem, io, dout);
}

int ds_omlsa_estimate(ds_handle* h, const float* y, const float* u, int n_frames, float* lambda_d, float* G, float* p, int mem) {
    if (!h || !y || !u || !lambda_d || !G || !p) return fail(h, DS_EINVAL, "ds_omlsa_estimate: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{y, u, nullptr}, {n * 4, n * (h->cfg.n_mics - 1) * 4, 0}, {lambda_d, G, p}, {n * 4, n * 4, n * 4}};
    return run_binop(h, DS_ALGO_OMLSA, "ds_omlsa_estimate", n_frames, mem, io, 0, 0);
}

int ds_omlsa_postfilter(ds_handle* h, const float* Y, const float* U, int n_frames, float* G, float* Yout, int mem) {
    if (!h || !Y || !U || !G || !Yout) return fail(h, DS_EINVAL, "ds_omlsa_postfilter: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    if (h->cfg.algo != DS_ALGO_OMLSA) return fail(h, DS_ESTATE, "ds_omlsa_postfilter: handle was created for a different algo");
    if (n_frames < 0) return fail(h, DS_ESHAPE, "ds_omlsa_postfilter: n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    rc = stage_reserve(h, 3, 2 * n * 4); if (rc) return rc;                     // lambda_d and p are not returned by this entry point
    IoSpec io = {{Y, U, nullptr}, {n * 8, n * (h->cfg.n_mics - 1) * 8, 0}, {nullptr, G, nullptr, Yout, nullptr}, {0, n * 4, 0, n * 8, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.KP = h->KP; p.T = n_frames; p.st = h->opst; p.NF = h->NF; p.M = h->cfg.n_mics;
    p.in0 = din[0]; p.in1 = din[1]; p.out0 = h->dev_buf[3]; p.out1 = dout[1]; p.out2 = h->dev_buf[3] + n; p.out3 = dout[3];
    p.frm_cnt = h->op_frm; p.ell = h->op_ell; p.L = h->mcra_L; p.first_frame = h->op_first; p.in_complex = 1; p.x_fan = 1;
    p.dev_cnt = h->use_dev_cnt ? h->dev_cnt : nullptr;
    take_tick(h, h->stream, p.tick);
    DS_HIP(h, ds::launch_binop(ds::OP_OMLSA, p, h->stream));
    if (h->use_dev_cnt) { rc = post_tick(h, h->dev_cnt, n_frames, h->mcra_L, 0, 0, h->stream); if (rc) return rc; }
    advance_host_counters(h, n_frames, h->mcra_L);
    return io_end(h, mem, io, dout);
}

int ds_sublms_update(ds_handle* h, const float* x, const float* d, const float* p, int n_frames, float* err, int mem) {
    if (!h || !x || !d || !err) return fail(h, DS_EINVAL, "ds_sublms_update: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{x, d, p}, {n * h->cfg.n_mics * 8, n * 8, p ? n * 4 : 0}, {err, nullptr, nullptr}, {n * 8, 0, 0}};
    return run_binop(h, DS_ALGO_SUBLMS, "ds_sublms_update", n_frames, mem, io, 0, p ? 1 : 0);
}

int ds_subrls_update(ds_handle* h, const float* x, const float* d, int n_frames, float* err, int mem) {
    if (!h || !x || !d || !err) return fail(h, DS_EINVAL, "ds_subrls_update: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{x, d, nullptr}, {n * 8, n * 8, 0}, {err, nullptr, nullptr}, {n * 8, 0, 0}};
    return run_binop(h, DS_ALGO_SUBRLS, "ds_subrls_update", n_frames, mem, io, 0, 0);
}


int ds_wpe_update(ds_handle* h, const float* x_delayed, const float* d, int n_frames, float* err, int mem) {
    if (!x_delayed) return fail(h, DS_EINVAL, "ds_wpe_update: NULL argument");
    return wpe_run(h, x_delayed, d, n_frames, err, mem, nullptr, 0, 0, nullptr);
}

}  // extern "C"
