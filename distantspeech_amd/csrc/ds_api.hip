// ds_api.hip — handle management and the C-ABI of libdsenh.so (see include/dsenh.h).
// No CPU compute path lives here: every ds_process* call launches the gfx950 kernels.
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
#include "ds_tables.hpp"

using ds::cf;
using ds::KernelInfo;
using ds::Params;

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
    int op;                     // ds::OP_* or -1
    float* opst;                // operator state [B][NF][KP]
    int NF;
    int op_frm, op_ell, op_first;   // uniform counters of the operator handle
    int filter_len, norm;
    float filt_mu, filt_alpha, rls_lambda;
    float* dev_buf[10];         // staging for host-pointer frame-level calls (3 in, 1 scratch, 5 out, 1 aux table)
    size_t dev_buf_bytes[10];
    size_t aux_floats;
    float* td_mem;              // DS_ALGO_FRONTEND: notch memories [B][M][2]
    float* td_cache[2];         // FIR history ping-pong [B][L-1][M]
    int td_L, td_cur;
    float* tdf_w; float* tdf_buf; float* tdf_P;     // DS_ALGO_TDNLMS / TDRLS state
    int fdaf_kind, fdaf_constrain, fdaf_non_causal, fdaf_weight_norm;   // DS_ALGO_FDAF (state lives in opst)
    int x_fan, p_complement;    // subband LMS / RLS inside a chain: shared reference input, 1 - p (OpParams)
    // DS_ALGO_WPE_MVDR: a chain of operator handles sharing this handle's stream, device-resident between the stages
    ds_handle* sub[10];         // WPE_MVDR: analysis transform, WPE, McMcra, adaptive frame loop, synthesis transform; SUBBAND_GSC: see chain2_*
    bool owns_stream;
    int wpe_delay;
    float* chain_buf[16];       // WPE_MVDR: D, -, E, p, G, Y, ring of the last wpe_delay analysis frames; SUBBAND_GSC: see chain2_reserve
    size_t chain_bytes[16];
    int hist_cur;               // ring slot of the oldest frame
    // cached hipGraph of a ds_process_device_seq() sequence
    hipGraphExec_t graph_exec;
    int split;                  // DS_PARAM_SPLIT: utterance groups captured as parallel graph branches
    hipStream_t side[7];        // side streams for the extra branches
    hipEvent_t ev_fork, ev_join[7];
    long long graph_key[16];
    bool graph_valid;
    float* x_stage;
    float* y_stage;
    size_t x_stage_elems, y_stage_elems;
    // params
    int method;
    int mcra_L;
    float alpha_y, alpha_v, diag, gate, mu, out_scale;
    std::string err;
};

namespace {

thread_local std::string g_err;

int fail(ds_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

#define DS_HIP(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, DS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

size_t bins_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * h->NP * h->KP * sizeof(ds::vec4); }
size_t tail_in_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * h->cfg.n_mics * h->cfg.hop * sizeof(float); }
size_t tail_out_bytes(const ds_handle* h) {
    const size_t ch = h->cfg.algo == DS_ALGO_TRANSFORM ? (size_t)h->cfg.n_mics : 1;   // Transform keeps one OLA tail per channel
    return (size_t)h->cfg.batch * ch * h->cfg.hop * sizeof(float);
}
size_t opst_bytes(const ds_handle* h) { return h->op >= 0 ? (size_t)h->cfg.batch * h->NF * h->KP * sizeof(float) : 0; }
size_t counters_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * 4 * sizeof(int); }

int set_device(ds_handle* h) {
    DS_HIP(h, hipSetDevice(h->device));
    return DS_OK;
}

int zero_state(ds_handle* h) {
    if (bins_bytes(h)) DS_HIP(h, hipMemsetAsync(h->bins, 0, bins_bytes(h), h->stream));
    DS_HIP(h, hipMemsetAsync(h->tail_in, 0, tail_in_bytes(h), h->stream));
    DS_HIP(h, hipMemsetAsync(h->tail_out, 0, tail_out_bytes(h), h->stream));
    // counters: mcra frm_cnt = 0, ell = 1 (NoiseEstimationBase.py:18,31), spp frm_cnt = 0
    std::vector<int> c((size_t)h->cfg.batch * 4, 0);
    for (int b = 0; b < h->cfg.batch; ++b) c[(size_t)b * 4 + 1] = 1;
    DS_HIP(h, hipMemcpyAsync(h->counters, c.data(), counters_bytes(h), hipMemcpyHostToDevice, h->stream));
    if (h->cfg.algo == DS_ALGO_TDNLMS || h->cfg.algo == DS_ALGO_TDRLS) {
        const size_t Lf = h->cfg.filter_len, Bt = h->cfg.batch;
        DS_HIP(h, hipMemsetAsync(h->tdf_w, 0, Bt * Lf * sizeof(float), h->stream));
        DS_HIP(h, hipMemsetAsync(h->tdf_buf, 0, Bt * Lf * sizeof(float), h->stream));
        if (h->tdf_P) {                                                        // RLS.py:20: P = eye / delta, delta = 1e-3
            std::vector<float> P0(Bt * Lf * Lf, 0.0f);
            for (size_t b = 0; b < Bt; ++b)
                for (size_t i = 0; i < Lf; ++i) P0[(b * Lf + i) * Lf + i] = 1000.0f;
            DS_HIP(h, hipMemcpyAsync(h->tdf_P, P0.data(), P0.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
            DS_HIP(h, hipStreamSynchronize(h->stream));
        }
    }
    if (h->cfg.algo == DS_ALGO_FRONTEND) {
        DS_HIP(h, hipMemsetAsync(h->td_mem, 0, (size_t)h->cfg.batch * h->cfg.n_mics * 2 * sizeof(float), h->stream));
        for (int i = 0; i < 2; ++i)
            if (h->td_cache[i]) DS_HIP(h, hipMemsetAsync(h->td_cache[i], 0, (size_t)h->cfg.batch * (h->td_L > 1 ? h->td_L - 1 : 1) * h->cfg.n_mics * sizeof(float), h->stream));
    }
    if (h->op >= 0 && h->NF > 0) {
        // operator state: zeros, except the rows the reference initialises to non-zero values
        std::vector<float> st((size_t)h->cfg.batch * h->NF * h->KP, 0.0f);
        auto fill_row = [&](int f, float v) {
            for (int b = 0; b < h->cfg.batch; ++b)
                for (int k = 0; k < h->KP; ++k) st[((size_t)b * h->NF + f) * h->KP + k] = v;
        };
        if (h->op == ds::OP_OMLSA) {                       // omlsa_multi.py:33-58: gamma, G_H1, G, xi_hat, q_hat = 1
            const int o_s = 5 * h->cfg.n_mics + 1 + (h->cfg.n_mics - 1);
            fill_row(o_s + 1, 1.0f); fill_row(o_s + 2, 1.0f); fill_row(o_s + 3, 1.0f); fill_row(o_s + 5, 1.0f); fill_row(o_s + 6, 1.0f);
            fill_row(5 * h->cfg.n_mics, 1.0f);             // zeta_Y = 1
        }
        if (h->op == ds::OP_WPE) {                         // awpe.py:69-73: P = I * 1e-3 (bin blocks of ds_wpe.hpp)
            const int C = h->cfg.n_mics, N = h->filter_len, CN = C * N, SB = ds::wpe_bin_floats(C, N);
            for (int b = 0; b < h->cfg.batch; ++b)
                for (int k = 0; k < h->K; ++k)
                    for (int i = 0; i < CN; ++i)
                        st[(size_t)b * h->NF * h->KP + (size_t)k * SB + 2 * (i * CN + i)] = 1e-3f;
        }
        if (h->op == ds::OP_SUBRLS) {                      // SubbandRLS.py:40-42: P = I / 1e-3
            const int N = h->filter_len;
            for (int i = 0; i < N; ++i) fill_row(4 * N + 2 * (i * N + i), 1000.0f);
        }
        DS_HIP(h, hipMemcpyAsync(h->opst, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        h->op_frm = 0; h->op_ell = 1; h->op_first = 1;
    }
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

void fill_params(const ds_handle* h, Params& p) {
    std::memset(&p, 0, sizeof(p));
    p.bins = h->bins;
    p.tail_in = h->tail_in;
    p.tail_out = h->tail_out;
    p.counters = h->counters;
    p.tables = h->tables;
    p.steer = h->steer;
    p.steer_batch_stride = h->steer_per_utt ? (long long)h->K * h->cfg.n_mics : 0;
    p.method = h->method;
    p.mcra_L = h->mcra_L;
    p.out_scale = h->out_scale;
    p.alpha_y = h->alpha_y;
    p.alpha_v = h->alpha_v;
    p.diag = h->diag;
    p.gate = h->gate;
    p.mu = h->mu;
}

}  // namespace

extern "C" {

int ds_version(void) { return DS_VERSION; }

int ds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ds_strerror(int code) {
    switch (code) {
        case DS_OK: return "ok";
        case DS_EINVAL: return "invalid argument";
        case DS_ESHAPE: return "shape mismatch (n_samples must be a multiple of hop)";
        case DS_EUNSUPPORTED: return "unsupported configuration (no compiled kernel)";
        case DS_EHIP: return "HIP runtime error";
        case DS_ENOMEM: return "out of memory";
        case DS_ESTATE: return "invalid call order";
        default: return "unknown error";
    }
}

const char* ds_last_error(const ds_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int ds_create(const ds_config* cfg, ds_handle** out) {
    if (!cfg || !out) return fail(nullptr, DS_EINVAL, "ds_create: NULL argument");
    if (cfg->struct_size != (int32_t)sizeof(ds_config)) return fail(nullptr, DS_EINVAL, "ds_create: struct_size mismatch");
    *out = nullptr;
    if (cfg->batch <= 0) return fail(nullptr, DS_EINVAL, "ds_create: batch must be > 0");
    if (cfg->algo <= DS_ALGO_TRANSFORM && cfg->hop * 2 != cfg->nfft)
        return fail(nullptr, DS_EUNSUPPORTED, "ds_create: only hop == nfft/2 is supported");
    KernelInfo ki = {nullptr, 0, 0, 0};
    KernelInfo ki_istft = {nullptr, 0, 0, 0};
    int op = -1, NF = 0;
    const int KPo = (cfg->nfft / 2 + 1 + 3) & ~3;
    const int flen = cfg->filter_len > 0 ? cfg->filter_len : 2;
    const bool is_tdf = cfg->algo == DS_ALGO_TDNLMS || cfg->algo == DS_ALGO_TDRLS;
    switch (cfg->algo) {
        case DS_ALGO_FIXED: ki = ds::lookup_fixed(cfg->nfft, cfg->n_mics); break;
        case DS_ALGO_ADAPTIVE:
            ki = cfg->track_ryy ? ds::lookup_adaptive_ryy(cfg->nfft, cfg->n_mics) : ds::lookup_adaptive_noryy(cfg->nfft, cfg->n_mics);
            break;
        case DS_ALGO_GSC: ki = ds::lookup_gsc(cfg->nfft, cfg->n_mics); break;
        case DS_ALGO_TRANSFORM:
            ki = ds::lookup_stft(cfg->nfft, cfg->n_mics);
            ki_istft = ds::lookup_istft(cfg->nfft, cfg->n_mics);
            break;
        case DS_ALGO_MCRA: op = ds::OP_MCRA; NF = 5; break;
        case DS_ALGO_MCMCRA:
            if (cfg->n_mics == 2 || cfg->n_mics == 4 || cfg->n_mics == 6 || cfg->n_mics == 8) {
                op = ds::OP_MCMCRA; NF = cfg->n_mics * (cfg->n_mics + 1) + 4;
            }
            break;
        case DS_ALGO_MCSPPBASE:
            if (cfg->n_mics == 2 || cfg->n_mics == 4 || cfg->n_mics == 6 || cfg->n_mics == 8) {
                op = ds::OP_MCSPPBASE; NF = ds::mcsppbase_nf(cfg->n_mics);
            }
            break;
        case DS_ALGO_MCSPP:
            if (ds::op_supported(ds::OP_MCSPP, cfg->n_mics) && cfg->n_mics >= 3) { op = ds::OP_MCSPP; NF = ds::mcspp_nf(cfg->n_mics); }
            break;
        case DS_ALGO_TDNLMS:
            if (cfg->filter_len >= 1 && cfg->filter_len <= ds::TDF_LMAX) { op = 101; NF = 0; }
            break;
        case DS_ALGO_TDRLS:
            if (cfg->filter_len >= 1 && cfg->filter_len <= ds::TDF_RLS_LMAX) { op = 102; NF = 0; }
            break;
        case DS_ALGO_FRONTEND:
            if (cfg->n_mics >= 1 && cfg->n_mics <= 16) { op = 100; NF = 0; }
            break;
        case DS_ALGO_ADAPTIVE_FRAMES:
            if (ds::op_supported(ds::OP_ADAPTIVE, cfg->n_mics)) { op = ds::OP_ADAPTIVE; NF = cfg->n_mics * cfg->n_mics + 5; }
            break;
        case DS_ALGO_SUBBAND_GSC:
            if (ds::op_supported(ds::OP_MCSPP, cfg->n_mics) && cfg->n_mics >= 3 && cfg->hop * 2 == cfg->nfft && flen <= ds::RLS_NMAX) { op = 105; NF = 0; }
            break;
        case DS_ALGO_WPE_MVDR:
            if (ds::op_supported(ds::OP_ADAPTIVE, cfg->n_mics) && cfg->hop * 2 == cfg->nfft && cfg->n_mics * flen <= ds::WPE_CNMAX) { op = 104; NF = 0; }
            break;
        case DS_ALGO_FDAF:
            if ((cfg->nfft == 128 || cfg->nfft == 256 || cfg->nfft == 512 || cfg->nfft == 1024) && cfg->n_mics >= 1 && cfg->n_mics <= 8) {
                op = 103;
                NF = (int)((ds::fdaf_state_floats(cfg->nfft, cfg->n_mics) + KPo - 1) / KPo);
            }
            break;
        case DS_ALGO_LINALG:
            if (ds::op_supported(ds::OP_STEERING, cfg->n_mics)) { op = ds::OP_STEERING; NF = 0; }
            break;
        case DS_ALGO_OMLSA:
            if (cfg->n_mics >= 2 && cfg->n_mics <= 16) { op = ds::OP_OMLSA; NF = ds::omlsa_nf(cfg->n_mics); }
            break;
        case DS_ALGO_SUBLMS:
            if (cfg->n_mics >= 1 && cfg->n_mics <= 16 && flen <= 8) { op = ds::OP_SUBLMS; NF = ds::sublms_nf(flen, cfg->n_mics); }
            break;
        case DS_ALGO_SUBRLS:
            if (flen <= ds::RLS_NMAX) { op = ds::OP_SUBRLS; NF = ds::subrls_nf(flen); }
            break;
        case DS_ALGO_WPE:
            if (cfg->n_mics >= 1 && cfg->n_mics <= ds::WPE_CMAX && cfg->n_mics * flen <= ds::WPE_CNMAX) {
                op = ds::OP_WPE;
                NF = (int)(((long long)(cfg->nfft / 2 + 1) * ds::wpe_bin_floats(cfg->n_mics, flen) + KPo - 1) / KPo);   // [B][K][bin block]
            }
            break;
        default: return fail(nullptr, DS_EINVAL, "ds_create: unknown algo");
    }
    if (op >= 0) {
        if (cfg->nfft < 4 || (cfg->nfft & 1)) return fail(nullptr, DS_EUNSUPPORTED, "ds_create: nfft must be even");
        ki.launch = nullptr; ki.NP = 0; ki.KP = KPo; ki.NT = 256;
    } else if (!ki.launch) {
        char buf[200];
        snprintf(buf, sizeof buf, "ds_create: no kernel for algo=%d nfft=%d n_mics=%d (nfft in {256,512,1024}, n_mics in {2,4,6,8}; Transform also n_mics=1)",
                 cfg->algo, cfg->nfft, cfg->n_mics);
        return fail(nullptr, DS_EUNSUPPORTED, buf);
    }
    if (cfg->algo >= DS_ALGO_MCRA && op < 0) return fail(nullptr, DS_EUNSUPPORTED, "ds_create: unsupported n_mics / filter_len for this frame-level object");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, DS_EHIP, "ds_create: no HIP device visible (libdsenh has no CPU path)");
    ds_handle* h = new (std::nothrow) ds_handle();
    if (!h) return fail(nullptr, DS_ENOMEM, "ds_create: host allocation failed");
    h->cfg = *cfg;
    h->ki = ki;
    h->K = cfg->nfft / 2 + 1;
    h->KP = ki.KP;
    h->NP = ki.NP;
    h->NT = ki.NT;
    h->bins = nullptr; h->tail_in = nullptr; h->tail_out = nullptr; h->counters = nullptr;
    h->tables = nullptr; h->steer = nullptr; h->x_stage = nullptr; h->y_stage = nullptr;
    h->x_stage_elems = h->y_stage_elems = 0;
    h->steer_per_utt = 0; h->steer_set = false;
    h->graph_exec = nullptr; h->graph_valid = false;
    h->split = 1; h->ev_fork = nullptr;
    for (int i = 0; i < 7; ++i) { h->side[i] = nullptr; h->ev_join[i] = nullptr; }
    h->ki_istft = ki_istft; h->op = op; h->opst = nullptr; h->NF = NF;
    h->op_frm = 0; h->op_ell = 1; h->op_first = 1;
    h->filter_len = flen; h->norm = cfg->no_norm ? 0 : 1;
    h->filt_mu = cfg->filt_mu > 0 ? cfg->filt_mu : (cfg->algo == DS_ALGO_SUBRLS ? 0.5f : 0.1f);
    h->filt_alpha = cfg->filt_alpha > 0 ? cfg->filt_alpha : 0.9f;
    h->rls_lambda = cfg->rls_lambda > 0 ? cfg->rls_lambda : 0.998f;
    for (int i = 0; i < 10; ++i) { h->dev_buf[i] = nullptr; h->dev_buf_bytes[i] = 0; }
    h->aux_floats = 0;
    h->tdf_w = h->tdf_buf = h->tdf_P = nullptr;
    h->x_fan = 1; h->p_complement = 0;
    h->fdaf_kind = DS_FDAF_PLAIN; h->fdaf_constrain = 1; h->fdaf_non_causal = 0; h->fdaf_weight_norm = 0;
    for (int i = 0; i < 10; ++i) h->sub[i] = nullptr;
    for (int i = 0; i < 16; ++i) { h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0; }
    h->owns_stream = true; h->wpe_delay = 4; h->hist_cur = 0;
    h->td_mem = nullptr; h->td_cache[0] = h->td_cache[1] = nullptr; h->td_L = 0; h->td_cur = 0;
    h->method = DS_METHOD_MVDR;
    h->mcra_L = cfg->mcra_L > 0 ? cfg->mcra_L : 15;
    h->alpha_y = cfg->alpha_y > 0 ? cfg->alpha_y : 0.8f;
    h->alpha_v = cfg->alpha_v > 0 ? cfg->alpha_v : 0.9998f;
    h->diag = cfg->diag > 0 ? cfg->diag : 1e-6f;
    h->gate = cfg->gate > 0 ? cfg->gate : 0.4f;
    h->mu = cfg->mu > 0 ? cfg->mu : 0.01f;
    if (cfg->device >= 0) h->device = cfg->device;
    else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    h->stream = nullptr; h->ev0 = nullptr; h->ev1 = nullptr;

#define DS_CRE(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            std::string m = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            ds_destroy(h);                                                                       \
            return fail(nullptr, DS_EHIP, m);                                                    \
        }                                                                                        \
    } while (0)

    DS_CRE(hipSetDevice(h->device));
    DS_CRE(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    DS_CRE(hipEventCreate(&h->ev0));
    DS_CRE(hipEventCreate(&h->ev1));
    if (bins_bytes(h)) DS_CRE(hipMalloc((void**)&h->bins, bins_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->tail_in, tail_in_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->tail_out, tail_out_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->counters, counters_bytes(h)));
    if (is_tdf) {
        const size_t Lf = cfg->filter_len;
        DS_CRE(hipMalloc((void**)&h->tdf_w, (size_t)cfg->batch * Lf * sizeof(float)));
        DS_CRE(hipMalloc((void**)&h->tdf_buf, (size_t)cfg->batch * Lf * sizeof(float)));
        if (cfg->algo == DS_ALGO_TDRLS) DS_CRE(hipMalloc((void**)&h->tdf_P, (size_t)cfg->batch * Lf * Lf * sizeof(float)));
        h->filt_mu = cfg->filt_mu > 0 ? cfg->filt_mu : (cfg->algo == DS_ALGO_TDRLS ? 0.5f : 0.1f);
        h->rls_lambda = cfg->rls_lambda > 0 ? cfg->rls_lambda : 0.9998f;                 // RLS.py:15
    }
    if (cfg->algo == DS_ALGO_FRONTEND) {
        DS_CRE(hipMalloc((void**)&h->td_mem, (size_t)cfg->batch * cfg->n_mics * 2 * sizeof(float)));
        DS_CRE(hipMemset(h->td_mem, 0, (size_t)cfg->batch * cfg->n_mics * 2 * sizeof(float)));
    }
    if (h->op >= 0 && h->NF > 0) DS_CRE(hipMalloc((void**)&h->opst, (size_t)cfg->batch * h->NF * h->KP * sizeof(float)));
    const int N = cfg->nfft, NC = N / 2;
    DS_CRE(hipMalloc((void**)&h->steer, (size_t)h->K * cfg->n_mics * sizeof(cf)));
    {
        std::vector<float> blob;
        ds::make_table_blob(N, cfg->hop, blob, h->out_scale);
        DS_CRE(hipMalloc((void**)&h->tables, blob.size() * sizeof(float)));
        DS_CRE(hipMemcpy(h->tables, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    }
#undef DS_CRE
    int rc = zero_state(h);
    if (rc != DS_OK) { std::string m = h->err; ds_destroy(h); return fail(nullptr, rc, m); }
    if (cfg->algo == DS_ALGO_WPE_MVDR) {
        const int algos[5] = {DS_ALGO_TRANSFORM, DS_ALGO_WPE, DS_ALGO_MCMCRA, DS_ALGO_ADAPTIVE_FRAMES, DS_ALGO_TRANSFORM};
        for (int i = 0; i < 5; ++i) {
            ds_config c = *cfg;
            c.algo = algos[i]; c.device = h->device;
            if (i == 4) c.n_mics = 1;
            rc = ds_create(&c, &h->sub[i]);
            if (rc != DS_OK) { std::string m = g_err; ds_destroy(h); return fail(nullptr, rc, "ds_create(DS_ALGO_WPE_MVDR): stage " + std::to_string(i) + ": " + m); }
            (void)hipStreamDestroy(h->sub[i]->stream);          // every stage runs on the chain's stream
            h->sub[i]->stream = h->stream; h->sub[i]->owns_stream = false;
        }
    }
    if (cfg->algo == DS_ALGO_SUBBAND_GSC) {
        // stages of SubbandGSC.__init__ (SubbandGSC.py:85-124): 0 front end (notch radius 0.98 + TimeAlignment), 1 transform (M),
        // 2 McSpp, 3 bm[m].transform_x (identical for all m: one 1-channel transform), 4 bm[m].transform_d (B * M 1-channel
        // transforms: analysis of the aligned channels and synthesis of the blocking-filter outputs), 5 the M blocking filters as one
        // batch of B * M, 6 aic_filter.transform_x (M), 7 aic_filter (SubbandLmsMc, mu 0.01, alpha 0.8), 8 aic_filter.transform_d
        const int M = cfg->n_mics;
        const bool rls = cfg->rls_lambda > 0.0f;
        for (int i = 0; i < 9; ++i) {
            ds_config c = *cfg;
            c.device = h->device; c.filter_len = flen;
            switch (i) {
                case 0: c.algo = DS_ALGO_FRONTEND; c.filt_alpha = 0.98f; break;
                case 1: case 6: c.algo = DS_ALGO_TRANSFORM; break;
                case 2: c.algo = DS_ALGO_MCSPP; break;
                case 3: case 8: c.algo = DS_ALGO_TRANSFORM; c.n_mics = 1; break;
                case 4: c.algo = DS_ALGO_TRANSFORM; c.n_mics = 1; c.batch = cfg->batch * M; break;
                case 5: c.algo = rls ? DS_ALGO_SUBRLS : DS_ALGO_SUBLMS; c.n_mics = 1; c.batch = cfg->batch * M;
                        c.filt_mu = rls ? 0.0f : 0.1f; c.filt_alpha = 0.0f; break;           // SubbandGSC.py:99-101 / SubbandRLS defaults
                case 7: c.algo = DS_ALGO_SUBLMS; c.filt_mu = 0.01f; c.filt_alpha = 0.8f; c.rls_lambda = 0.0f; break;   // :103-109
            }
            rc = ds_create(&c, &h->sub[i]);
            if (rc != DS_OK) { std::string m = g_err; ds_destroy(h); return fail(nullptr, rc, "ds_create(DS_ALGO_SUBBAND_GSC): stage " + std::to_string(i) + ": " + m); }
            (void)hipStreamDestroy(h->sub[i]->stream);
            h->sub[i]->stream = h->stream; h->sub[i]->owns_stream = false;
        }
        h->sub[5]->x_fan = M;                 // the M blocking filters of an utterance share its fixed-beamformer spectrum and p
        h->sub[7]->p_complement = 1;          // SubbandGSC.py:232: p = 1 - p
    }
    *out = h;
    return DS_OK;
}

int ds_destroy(ds_handle* h) {
    if (!h) return DS_EINVAL;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 10; ++i) if (h->sub[i]) (void)ds_destroy(h->sub[i]);
    for (int i = 0; i < 16; ++i) (void)hipFree(h->chain_buf[i]);
    (void)hipFree(h->bins); (void)hipFree(h->tail_in); (void)hipFree(h->tail_out); (void)hipFree(h->counters);
    (void)hipFree(h->tables); (void)hipFree(h->steer);
    (void)hipFree(h->x_stage); (void)hipFree(h->y_stage); (void)hipFree(h->opst);
    for (int i = 0; i < 10; ++i) (void)hipFree(h->dev_buf[i]);
    (void)hipFree(h->tdf_w); (void)hipFree(h->tdf_buf); (void)hipFree(h->tdf_P);
    (void)hipFree(h->td_mem); (void)hipFree(h->td_cache[0]); (void)hipFree(h->td_cache[1]);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    for (int i = 0; i < 7; ++i) { if (h->side[i]) (void)hipStreamDestroy(h->side[i]); if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream && h->owns_stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return DS_OK;
}

int ds_reset(ds_handle* h) {
    if (!h) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    for (int i = 0; i < 10; ++i)
        if (h->sub[i]) { rc = ds_reset(h->sub[i]); if (rc) return fail(h, rc, h->sub[i]->err); }
    if (h->cfg.algo == DS_ALGO_WPE_MVDR && h->chain_buf[6]) DS_HIP(h, hipMemsetAsync(h->chain_buf[6], 0, h->chain_bytes[6], h->stream));
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC)
        for (int i = 13; i < 15; ++i)
            if (h->chain_buf[i]) DS_HIP(h, hipMemsetAsync(h->chain_buf[i], 0, h->chain_bytes[i], h->stream));
    h->hist_cur = 0;
    return zero_state(h);
}

int ds_set_steering(ds_handle* h, const float* steer, int per_utterance) {
    if (!h || !steer) return fail(h, DS_EINVAL, "ds_set_steering: NULL argument");
    int rc = set_device(h);
    if (rc) return rc;
    const size_t one = (size_t)h->K * h->cfg.n_mics * sizeof(cf);
    const size_t need = per_utterance ? one * h->cfg.batch : one;
    if ((per_utterance != 0) != (h->steer_per_utt != 0)) {
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->steer);
        h->steer = nullptr;
        DS_HIP(h, hipMalloc((void**)&h->steer, need));
        h->steer_per_utt = per_utterance ? 1 : 0;
    }
    DS_HIP(h, hipMemcpyAsync(h->steer, steer, need, hipMemcpyHostToDevice, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    h->steer_set = true;
    if (h->sub[3]) { rc = ds_set_steering(h->sub[3], steer, per_utterance); if (rc) return fail(h, rc, h->sub[3]->err); }
    return DS_OK;
}

int ds_set_param_i(ds_handle* h, int id, int value) {
    if (!h) return DS_EINVAL;
    if (h->sub[3] && (id == DS_PARAM_METHOD || id == DS_PARAM_MCRA_L)) {
        const int rc = ds_set_param_i(h->sub[3], id, value);
        if (rc) return fail(h, rc, h->sub[3]->err);
    }
    switch (id) {
        case DS_PARAM_WPE_DELAY:
            if (h->cfg.algo != DS_ALGO_WPE_MVDR || value < 0 || value > 64) return fail(h, DS_EINVAL, "wpe delay: chain handles only, 0..64 frames");
            if (h->chain_buf[6]) return fail(h, DS_ESTATE, "wpe delay must be set before the first call");
            h->wpe_delay = value;
            return DS_OK;
        case DS_PARAM_METHOD:
            if (value < 0 || value > 3) return fail(h, DS_EINVAL, "method must be 0..3");
            if (value == DS_METHOD_TFGSC && h->cfg.algo == DS_ALGO_ADAPTIVE && !h->cfg.track_ryy)
                return fail(h, DS_ESTATE, "method TFGSC needs ds_config.track_ryy = 1");
            h->method = value;
            return DS_OK;
        case DS_PARAM_MCRA_L:
            if (value <= 0) return fail(h, DS_EINVAL, "mcra_L must be > 0");
            h->mcra_L = value;
            return DS_OK;
        case DS_PARAM_FDAF_KIND:
            if (value < DS_FDAF_PLAIN || value > DS_FDAF_AIC) return fail(h, DS_EINVAL, "FDAF kind must be 0..2");
            h->fdaf_kind = value;
            return DS_OK;
        case DS_PARAM_FDAF_CONSTRAIN: h->fdaf_constrain = value != 0; return DS_OK;
        case DS_PARAM_FDAF_NON_CAUSAL: h->fdaf_non_causal = value != 0; return DS_OK;
        case DS_PARAM_FDAF_WEIGHT_NORM: h->fdaf_weight_norm = value != 0; return DS_OK;
        case DS_PARAM_SPLIT:
            if (value < 1 || value > 8) return fail(h, DS_EINVAL, "split must be 1..8");
            h->split = value; h->graph_valid = false;
            return DS_OK;
        default: return fail(h, DS_EINVAL, "unknown int parameter id");
    }
}

int ds_set_param_f(ds_handle* h, int id, float value) {
    if (!h) return DS_EINVAL;
    if (h->sub[3] && (id == DS_PARAM_ALPHA_V || id == DS_PARAM_DIAG || id == DS_PARAM_GATE)) (void)ds_set_param_f(h->sub[3], id, value);
    switch (id) {
        case DS_PARAM_ALPHA_Y: h->alpha_y = value; return DS_OK;
        case DS_PARAM_ALPHA_V: h->alpha_v = value; return DS_OK;
        case DS_PARAM_DIAG: h->diag = value; return DS_OK;
        case DS_PARAM_GATE: h->gate = value; return DS_OK;
        case DS_PARAM_MU: h->mu = value; return DS_OK;
        default: return fail(h, DS_EINVAL, "unknown float parameter id");
    }
}

static int chain_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                                int n_samples, float* y_dev, long long y_batch_stride);
static int chain2_run(ds_handle* h, const float* x_dev, long long x_bstride, long long x_cstride, int n, float* y_dev, long long y_bstride,
                      float* fix_dev, float* bm_dev, float* p_dev, float* al_dev);

int ds_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                      int n_samples, float* y_dev, long long y_batch_stride, int first, int count, void* stream) {
    if (!h || !x_dev || !y_dev) return fail(h, DS_EINVAL, "ds_process_device: NULL argument");
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC) {
        if (first != 0 || count != h->cfg.batch) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle processes its whole batch");
        if (stream && (hipStream_t)stream != h->stream) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle runs on its own stream (pass NULL)");
        if (layout != DS_LAYOUT_CHANNELS_SAMPLES) return fail(h, DS_EUNSUPPORTED, "ds_process_device: the SubbandGSC chain takes [B][M][n] input");
        if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
        if (n_samples == 0) return DS_OK;
        return chain2_run(h, x_dev, x_batch_stride, x_chan_stride > 0 ? x_chan_stride : n_samples, n_samples, y_dev, y_batch_stride,
                          nullptr, nullptr, nullptr, nullptr);
    }
    if (h->cfg.algo == DS_ALGO_WPE_MVDR) {
        if (first != 0 || count != h->cfg.batch) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle processes its whole batch");
        if (stream && (hipStream_t)stream != h->stream) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle runs on its own stream (pass NULL)");
        if (!h->steer_set) return fail(h, DS_ESTATE, "ds_process_device: call ds_set_steering first");
        if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
        if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES) return fail(h, DS_EINVAL, "ds_process_device: unknown layout");
        if (n_samples == 0) return DS_OK;
        return chain_process_device(h, x_dev, layout, x_batch_stride, x_chan_stride, n_samples, y_dev, y_batch_stride);
    }
    if (h->cfg.algo > DS_ALGO_GSC) return fail(h, DS_ESTATE, "ds_process_device: this handle is a frame-level object; use ds_stft / ds_*_estimate / ds_sub*_update");
    if (!h->steer_set) return fail(h, DS_ESTATE, "ds_process_device: call ds_set_steering first");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0)
        return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
    if (first < 0 || count < 0 || first + count > h->cfg.batch)
        return fail(h, DS_EINVAL, "ds_process_device: utterance range outside the handle's batch");
    if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES)
        return fail(h, DS_EINVAL, "ds_process_device: unknown layout");
    if (((uintptr_t)x_dev & 15) || ((uintptr_t)y_dev & 15) || (x_batch_stride & 3) || (x_chan_stride & 3))
        return fail(h, DS_EINVAL, "ds_process_device: device buffers must be 16-byte aligned");
    if (n_samples == 0 || count == 0) return DS_OK;
    int rc = set_device(h);
    if (rc) return rc;
    Params p;
    fill_params(h, p);
    p.x = x_dev;
    p.y = y_dev;
    p.x_batch_stride = x_batch_stride;
    p.y_batch_stride = y_batch_stride;
    if (layout == DS_LAYOUT_CHANNELS_SAMPLES) { p.x_sample_stride = 1; p.x_chan_stride = x_chan_stride > 0 ? x_chan_stride : n_samples; }
    else { p.x_sample_stride = h->cfg.n_mics; p.x_chan_stride = 1; }
    p.T = n_samples / h->cfg.hop;
    p.batch0 = first;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    DS_HIP(h, h->ki.launch(p, count, s));
    return DS_OK;
}

int ds_process_device_seq(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                          long long x_call_stride, int n_samples_per_call, int n_calls, float* y_dev,
                          long long y_batch_stride, long long y_call_stride, int first, int count, void* stream,
                          int graph) {
    if (!h) return DS_EINVAL;
    if (n_calls < 0 || (x_call_stride & 3) || (y_call_stride & 3))
        return fail(h, DS_EINVAL, "ds_process_device_seq: bad n_calls / call strides (must be multiples of 4 elements)");
    if (n_calls == 0) return DS_OK;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (graph != 0 && (h->cfg.algo == DS_ALGO_WPE_MVDR || h->cfg.algo == DS_ALGO_SUBBAND_GSC))
        return fail(h, DS_EUNSUPPORTED, "ds_process_device_seq: chain handles keep frame counters on the host; use graph = 0");
    if (graph == 0) {
        for (int i = 0; i < n_calls; ++i) {
            int rc = ds_process_device(h, x_dev + (long long)i * x_call_stride, layout, x_batch_stride, x_chan_stride,
                                       n_samples_per_call, y_dev + (long long)i * y_call_stride, y_batch_stride, first,
                                       count, (void*)s);
            if (rc) return rc;
        }
        return DS_OK;
    }
    int rc = set_device(h);
    if (rc) return rc;
    float fl[6] = {h->alpha_y, h->alpha_v, h->diag, h->gate, h->mu, 0.0f};
    long long fbits[3];
    std::memcpy(fbits, fl, sizeof fbits);
    const long long key[16] = {(long long)(uintptr_t)x_dev, (long long)(uintptr_t)y_dev, layout, x_batch_stride, x_chan_stride,
                               x_call_stride, n_samples_per_call, n_calls, y_batch_stride, y_call_stride,
                               ((long long)first << 32) | (unsigned)count, ((long long)h->method << 32) | (unsigned)h->mcra_L,
                               fbits[0], fbits[1], fbits[2] ^ ((long long)h->split << 40), (long long)(uintptr_t)h->steer};
    if (!h->graph_valid || std::memcmp(key, h->graph_key, sizeof key) != 0) {
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
        h->graph_valid = false;
        hipStream_t cs = h->stream;                       // capture on the handle's own stream
        DS_HIP(h, hipStreamSynchronize(cs));
        const int ns = h->split < count ? h->split : (count > 0 ? count : 1);
        if (ns > 1) {
            if (!h->ev_fork) DS_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            for (int i = 0; i < ns - 1; ++i) {
                if (!h->side[i]) DS_HIP(h, hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking));
                if (!h->ev_join[i]) DS_HIP(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
            }
        }
        DS_HIP(h, hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int crc = DS_OK;
        if (ns > 1) {
            // independent utterance groups become parallel branches of the graph: while one group's kernels are in
            // their state-load / store phases the other group's kernels compute (utterances never interact)
            hipError_t e1 = hipEventRecord(h->ev_fork, cs);
            for (int g2 = 0; g2 < ns - 1 && e1 == hipSuccess; ++g2) e1 = hipStreamWaitEvent(h->side[g2], h->ev_fork, 0);
            if (e1 != hipSuccess) crc = DS_EHIP;
        }
        for (int g2 = 0; g2 < ns && crc == DS_OK; ++g2) {
            const int lo = (int)((long long)count * g2 / ns), hi = (int)((long long)count * (g2 + 1) / ns);
            hipStream_t bs = g2 == 0 ? cs : h->side[g2 - 1];
            for (int i = 0; i < n_calls && crc == DS_OK; ++i)
                crc = ds_process_device(h, x_dev + (long long)i * x_call_stride + (long long)lo * x_batch_stride, layout, x_batch_stride,
                                        x_chan_stride, n_samples_per_call, y_dev + (long long)i * y_call_stride + (long long)lo * y_batch_stride,
                                        y_batch_stride, first + lo, hi - lo, (void*)bs);
            if (g2 > 0 && crc == DS_OK) {
                if (hipEventRecord(h->ev_join[g2 - 1], bs) != hipSuccess || hipStreamWaitEvent(cs, h->ev_join[g2 - 1], 0) != hipSuccess) crc = DS_EHIP;
            }
        }
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(cs, &g);
        if (crc != DS_OK) { if (g) (void)hipGraphDestroy(g); return crc; }
        if (e != hipSuccess) return fail(h, DS_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(h, DS_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        std::memcpy(h->graph_key, key, sizeof key);
        h->graph_valid = true;
    }
    if (graph == 2) return DS_OK;
    DS_HIP(h, hipGraphLaunch(h->graph_exec, s));
    return DS_OK;
}

int ds_process(ds_handle* h, const float* x, int layout, int n_samples, float* y) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_process: NULL argument");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0)
        return fail(h, DS_ESHAPE, "ds_process: n_samples must be a multiple of hop");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h);
    if (rc) return rc;
    const size_t B = (size_t)h->cfg.batch, M = (size_t)h->cfg.n_mics;
    const size_t xe = B * M * (size_t)n_samples, ye = B * (size_t)n_samples;
    if (xe > h->x_stage_elems) {
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->x_stage); h->x_stage = nullptr; h->x_stage_elems = 0;
        DS_HIP(h, hipMalloc((void**)&h->x_stage, xe * sizeof(float)));
        h->x_stage_elems = xe;
    }
    if (ye > h->y_stage_elems) {
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->y_stage); h->y_stage = nullptr; h->y_stage_elems = 0;
        DS_HIP(h, hipMalloc((void**)&h->y_stage, ye * sizeof(float)));
        h->y_stage_elems = ye;
    }
    DS_HIP(h, hipMemcpyAsync(h->x_stage, x, xe * sizeof(float), hipMemcpyHostToDevice, h->stream));
    rc = ds_process_device(h, h->x_stage, layout, (long long)(M * (size_t)n_samples), 0, n_samples, h->y_stage,
                           (long long)n_samples, 0, h->cfg.batch, nullptr);
    if (rc) return rc;
    DS_HIP(h, hipMemcpyAsync(y, h->y_stage, ye * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

// ---- frame-level entry points ---------------------------------------------------------------------
namespace {

// make sure staging slot `i` holds at least `bytes`
int stage_reserve(ds_handle* h, int i, size_t bytes) {
    if (bytes <= h->dev_buf_bytes[i]) return DS_OK;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->dev_buf[i]); h->dev_buf[i] = nullptr; h->dev_buf_bytes[i] = 0;
    DS_HIP(h, hipMalloc((void**)&h->dev_buf[i], bytes));
    h->dev_buf_bytes[i] = bytes;
    return DS_OK;
}

struct IoSpec { const float* in[3]; size_t in_bytes[3]; float* out[5]; size_t out_bytes[5]; };

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

int run_binop(ds_handle* h, int want_algo, const char* who, int n_frames, int mem, const IoSpec& io, int is_complex, int has_p) {
    if (!h) return DS_EINVAL;
    if (h->cfg.algo != want_algo) return fail(h, DS_ESTATE, std::string(who) + ": handle was created for a different algo");
    if (n_frames < 0) return fail(h, DS_ESHAPE, std::string(who) + ": n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.KP = h->KP; p.T = n_frames;
    p.st = h->opst; p.NF = h->NF;
    p.in0 = din[0]; p.in1 = din[1]; p.in2 = din[2];
    p.out0 = dout[0]; p.out1 = dout[1]; p.out2 = dout[2];
    p.M = h->cfg.n_mics; p.N = h->filter_len;
    p.frm_cnt = h->op_frm; p.ell = h->op_ell; p.L = h->mcra_L; p.first_frame = h->op_first;
    p.in_complex = is_complex; p.has_p = has_p; p.norm = h->norm;
    p.mu = h->filt_mu; p.alpha = h->filt_alpha; p.reg = 1e-4f; p.lam = h->rls_lambda;
    p.x_fan = h->x_fan > 0 ? h->x_fan : 1; p.p_complement = h->p_complement;
    p.steer = h->steer; p.steer_batch_stride = h->steer_per_utt ? (long long)h->K * h->cfg.n_mics : 0;
    p.method = h->method; p.alpha_v = h->alpha_v; p.gate = h->gate; p.diag = h->diag;
    DS_HIP(h, ds::launch_binop(h->op, p, h->stream));
    // advance the uniform counters exactly like the kernel did (mcra.py:52-56,72-74)
    for (int t = 0; t < n_frames; ++t) {
        if (h->op_frm != 0 && h->op_ell % h->mcra_L == 0) h->op_ell = 0;
        h->op_frm += 1; h->op_ell += 1;
    }
    h->op_first = 0;
    return io_end(h, mem, io, dout);
}

}  // namespace

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
    DS_HIP(h, h->ki.launch(p, h->cfg.batch, h->stream));
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
    DS_HIP(h, h->ki_istft.launch(p, h->cfg.batch, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_mcra_estimate(ds_handle* h, const float* Y, int is_complex, int n_frames, float* lambda_d, int mem) {
    if (!h || !Y || !lambda_d) return fail(h, DS_EINVAL, "ds_mcra_estimate: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{Y, nullptr, nullptr}, {n * (is_complex ? 8 : 4), 0, 0}, {lambda_d, nullptr, nullptr}, {n * 4, 0, 0}};
    return run_binop(h, DS_ALGO_MCRA, "ds_mcra_estimate", n_frames, mem, io, is_complex ? 1 : 0, 0);
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
    return DS_OK;
}

int ds_mcspp_estimate(ds_handle* h, const float* y, int n_frames, float* p_out, float* w_pmwf, float* yout, float* phi_xx,
                      float* phi_vv_inv, int mem) {
    if (!h || !y || !p_out || !w_pmwf) return fail(h, DS_EINVAL, "ds_mcspp_estimate: NULL argument");
    if (h->cfg.algo != DS_ALGO_MCSPP) return fail(h, DS_ESTATE, "ds_mcspp_estimate: handle was created for a different algo");
    if ((phi_xx == nullptr) != (phi_vv_inv == nullptr)) return fail(h, DS_EINVAL, "ds_mcspp_estimate: phi_xx and phi_vv_inv go together");
    if (h->aux_floats < (size_t)h->K) return fail(h, DS_ESTATE, "ds_mcspp_estimate: call ds_set_aux(h, Fn[K]) first");
    if (n_frames < 0) return fail(h, DS_ESHAPE, "ds_mcspp_estimate: n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_frames * h->K, M = h->cfg.n_mics;
    IoSpec io = {{y, nullptr, nullptr}, {n * M * 8, 0, 0}, {p_out, w_pmwf, yout, phi_xx, phi_vv_inv},
                 {n * 4, n * M * 8, yout ? n * 8 : 0, phi_xx ? n * M * M * 8 : 0, phi_vv_inv ? n * M * M * 8 : 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    const size_t nbt = (size_t)h->cfg.batch * n_frames;
    rc = stage_reserve(h, 3, (n + nbt) * 4); if (rc) return rc;              // Gamma [B][T][K], then its band mean [B][T]
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.KP = h->KP; p.T = n_frames; p.st = h->opst; p.NF = h->NF; p.M = h->cfg.n_mics;
    p.frm_cnt = h->op_frm; p.ell = h->op_ell; p.L = 65;                      // mccdr.py:60-61
    p.in0 = din[0]; p.in1 = h->dev_buf[9]; p.out0 = h->dev_buf[3];
    DS_HIP(h, ds::launch_binop(ds::OP_MCCDR, p, h->stream));
    DS_HIP(h, ds::launch_mcspp_qavg(h->dev_buf[3], h->dev_buf[3] + n, (int)nbt, h->K, h->stream));
    p.in1 = h->dev_buf[3]; p.in2 = h->dev_buf[3] + n; p.N = 9;
    p.out0 = dout[0]; p.out1 = dout[1]; p.out2 = dout[2]; p.out3 = dout[3]; p.out4 = dout[4];
    DS_HIP(h, ds::launch_binop(ds::OP_MCSPP, p, h->stream));
    for (int t = 0; t < n_frames; ++t) {
        if (h->op_frm != 0 && h->op_ell % 65 == 0) h->op_ell = 0;
        h->op_frm += 1; h->op_ell += 1;
    }
    return io_end(h, mem, io, dout);
}

static int run_linalg(ds_handle* h, int op, const char* who, const IoSpec& io, int mem) {
    if (h->cfg.algo != DS_ALGO_LINALG) return fail(h, DS_ESTATE, std::string(who) + ": handle is not a DS_ALGO_LINALG object");
    int rc = set_device(h); if (rc) return rc;
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.KP = h->KP; p.T = 1; p.M = h->cfg.n_mics;
    p.in0 = din[0]; p.in1 = din[1]; p.out0 = dout[0];
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
    p.radius = h->cfg.filt_alpha > 0 ? h->cfg.filt_alpha : 0.9f;
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
    if (h->td_L != Lt) {                                                      // (re)allocate the history for this tap count
        DS_HIP(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < 2; ++i) {
            (void)hipFree(h->td_cache[i]); h->td_cache[i] = nullptr;
            const size_t cb = (size_t)h->cfg.batch * (Lt > 1 ? Lt - 1 : 1) * M * sizeof(float);
            DS_HIP(h, hipMalloc((void**)&h->td_cache[i], cb));
            DS_HIP(h, hipMemset(h->td_cache[i], 0, cb));
        }
        h->td_L = Lt; h->td_cur = 0;
    }
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
    if (p_mode < DS_FDAF_P_NONE || p_mode > DS_FDAF_P_BIN || (p_mode != DS_FDAF_P_NONE && !pp))
        return fail(h, DS_EINVAL, "ds_fdaf_update: p_mode / p mismatch");
    const int L = h->cfg.nfft / 2, C = h->cfg.n_mics;
    if (fir_truncate > L) return fail(h, DS_ESHAPE, "ds_fdaf_update: fir_truncate > filter_len");
    if (n_blocks == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_blocks * L;
    const size_t pbytes = p_mode == DS_FDAF_P_NONE ? 0 : (size_t)h->cfg.batch * n_blocks * (p_mode == DS_FDAF_P_BIN ? h->K : 1) * 4;
    IoSpec io = {{x, d, p_mode == DS_FDAF_P_NONE ? nullptr : pp}, {n * C * 4, n * 4, pbytes},
                 {err, w_out, nullptr, nullptr, nullptr}, {n * 4, w_out ? (size_t)h->cfg.batch * L * C * 4 : 0, 0, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::FdafParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.T = n_blocks; p.C = C;
    p.kind = h->fdaf_kind; p.constrain = h->fdaf_constrain; p.non_causal = h->fdaf_non_causal; p.weight_norm = h->fdaf_weight_norm;
    p.trunc = fir_truncate < 0 ? -1 : fir_truncate; p.p_mode = p_mode;
    p.mu = h->filt_mu; p.alpha = h->filt_alpha;
    p.x = din[0]; p.d = din[1]; p.p = din[2]; p.err = dout[0]; p.w_out = w_out ? dout[1] : nullptr;
    p.state = h->opst; p.state_stride = (long long)h->NF * h->KP;
    p.tables = h->tables;
    DS_HIP(h, ds::launch_fdaf(p, h->cfg.nfft, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_omlsa_estimate(ds_handle* h, const float* y, const float* u, int n_frames, float* lambda_d, float* G, float* p, int mem) {
    if (!h || !y || !u || !lambda_d || !G || !p) return fail(h, DS_EINVAL, "ds_omlsa_estimate: NULL argument");
    const size_t n = (size_t)h->cfg.batch * (n_frames > 0 ? n_frames : 0) * h->K;
    IoSpec io = {{y, u, nullptr}, {n * 4, n * (h->cfg.n_mics - 1) * 4, 0}, {lambda_d, G, p}, {n * 4, n * 4, n * 4}};
    return run_binop(h, DS_ALGO_OMLSA, "ds_omlsa_estimate", n_frames, mem, io, 0, 0);
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

static int wpe_run(ds_handle* h, const float* x_delayed, const float* d, int n_frames, float* err, int mem, float* ring, int ring_pos, int ring_len) {
    if (!h || (!x_delayed && !ring) || !d || !err) return fail(h, DS_EINVAL, "ds_wpe_update: NULL argument");
    if (h->cfg.algo != DS_ALGO_WPE) return fail(h, DS_ESTATE, "ds_wpe_update: handle was created for a different algo");
    if (n_frames < 0) return fail(h, DS_ESHAPE, "ds_wpe_update: n_frames < 0");
    if (n_frames == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t n = (size_t)h->cfg.batch * n_frames * h->K * h->cfg.n_mics * 8;
    IoSpec io = {{x_delayed, d, nullptr}, {x_delayed ? n : 0, n, 0}, {err, nullptr, nullptr}, {n, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    ds::WpeParams p;
    std::memset(&p, 0, sizeof p);
    p.B = h->cfg.batch; p.K = h->K; p.T = n_frames; p.C = h->cfg.n_mics; p.N = h->filter_len;
    p.xd = din[0]; p.d = din[1]; p.err = dout[0]; p.state = h->opst; p.lam = h->rls_lambda;
    p.ustride = (long long)h->NF * h->KP;
    p.ring = ring; p.ring_pos = ring_pos; p.ring_len = ring_len;
    DS_HIP(h, ds::launch_wpe(p, h->stream));
    return io_end(h, mem, io, dout);
}

int ds_wpe_update(ds_handle* h, const float* x_delayed, const float* d, int n_frames, float* err, int mem) {
    if (!x_delayed) return fail(h, DS_EINVAL, "ds_wpe_update: NULL argument");
    return wpe_run(h, x_delayed, d, n_frames, err, mem, nullptr, 0, 0);
}

// DS_ALGO_WPE_MVDR: STFT -> frame delay line -> WPE -> McMcra gain -> adaptive MVDR frame loop x gain -> ISTFT, every stage a
// kernel on h->stream reading the previous stage's device buffer (nothing returns to the host between the stages)
static int chain_reserve(ds_handle* h, int T) {
    const size_t B = h->cfg.batch, K = h->K, M = h->cfg.n_mics, d = h->wpe_delay > 0 ? h->wpe_delay : 1;
    const size_t need[8] = {B * T * K * M * 8, 0, B * T * K * M * 8, B * T * K * 4, B * T * K * 4, B * T * K * 8, B * d * K * M * 8, 0};
    for (int i = 0; i < 8; ++i) {
        if (need[i] == 0 || need[i] <= h->chain_bytes[i]) continue;
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->chain_buf[i]); h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0;
        DS_HIP(h, hipMalloc((void**)&h->chain_buf[i], need[i]));
        h->chain_bytes[i] = need[i];
        if (i >= 6) DS_HIP(h, hipMemset(h->chain_buf[i], 0, need[i]));      // the stream starts from silence (DelaySamples, awpe.py:75-76)
    }
    return DS_OK;
}

static int chain_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                                int n_samples, float* y_dev, long long y_batch_stride) {
    int rc = set_device(h); if (rc) return rc;
    const int B = h->cfg.batch, M = h->cfg.n_mics, K = h->K, T = n_samples / h->cfg.hop;
    rc = chain_reserve(h, T); if (rc) return rc;
    float *D = h->chain_buf[0], *E = h->chain_buf[2], *pp = h->chain_buf[3], *G = h->chain_buf[4], *Y = h->chain_buf[5];
#define DS_SUB(i, call) do { int rc_ = (call); if (rc_) return fail(h, rc_, h->sub[i]->err); } while (0)
    {   // analysis, strided input like the fused kernels take it
        ds_handle* t = h->sub[0];
        Params p;
        fill_params(t, p);
        p.x = x_dev; p.y = D;
        p.x_batch_stride = x_batch_stride;
        p.y_batch_stride = (long long)T * K * M * 2;
        if (layout == DS_LAYOUT_CHANNELS_SAMPLES) { p.x_sample_stride = 1; p.x_chan_stride = x_chan_stride > 0 ? x_chan_stride : n_samples; }
        else { p.x_sample_stride = M; p.x_chan_stride = 1; }
        p.T = T; p.batch0 = 0;
        DS_HIP(h, t->ki.launch(p, B, h->stream));
    }
    // delayed input of the prediction filter: a ring of the last wpe_delay analysis frames kept by the WPE kernel itself
    if (h->wpe_delay > 0) {
        DS_SUB(1, wpe_run(h->sub[1], nullptr, D, T, E, DS_MEM_DEVICE, h->chain_buf[6], h->hist_cur, h->wpe_delay));
        h->hist_cur = (h->hist_cur + T) % h->wpe_delay;
    } else {
        DS_SUB(1, wpe_run(h->sub[1], D, D, T, E, DS_MEM_DEVICE, nullptr, 0, 0));
    }
    DS_SUB(2, ds_mcmcra_estimate(h->sub[2], E, T, pp, G, DS_MEM_DEVICE));
    DS_SUB(3, ds_adaptive_frames(h->sub[3], E, G, T, Y, DS_MEM_DEVICE));
    {   // synthesis straight into the caller's (strided) output
        ds_handle* t = h->sub[4];
        Params p;
        fill_params(t, p);
        p.x = Y; p.y = y_dev;
        p.x_batch_stride = (long long)T * K * 2;
        p.y_batch_stride = y_batch_stride;
        p.T = T; p.batch0 = 0; p.method = 1;
        DS_HIP(h, t->ki_istft.launch(p, B, h->stream));
    }
#undef DS_SUB
    return DS_OK;
}

// ---- DS_ALGO_SUBBAND_GSC: SubbandGSC.process (SubbandGSC.py:170-262) as a device-resident chain ------------------------------
// buffers: 0 xn [B][M][n] (notched), 1 xa [B][M][n] (aligned), 2 fixed [B][n], 3 D c[B][T][K][M], 4 p [B][T][K], 5 PMWF scratch,
// 6 F c[B][T][K], 7 Dm c[B*M][T][K], 8 E c[B*M][T][K], 9 bm_td [B][M][n], 10 Xa c[B][T][K][M], 11 Dd c[B][T][K], 12 e2 c[B][T][K],
// 13 F of the previous block c[B][K] (state), 14 fixed output of the previous block [B][hop] (state)
static int chain2_reserve(ds_handle* h, int n) {
    const size_t B = h->cfg.batch, K = h->K, M = h->cfg.n_mics, T = n / h->cfg.hop, hop = h->cfg.hop;
    const size_t need[15] = {B * M * n * 4, B * M * n * 4, B * n * 4, B * T * K * M * 8, B * T * K * 4, B * T * K * M * 8, B * T * K * 8,
                             B * M * T * K * 8, B * M * T * K * 8, B * M * n * 4, B * T * K * M * 8, B * T * K * 8, B * T * K * 8,
                             B * K * 8, B * hop * 4};
    for (int i = 0; i < 15; ++i) {
        if (need[i] <= h->chain_bytes[i]) continue;
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->chain_buf[i]); h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0;
        DS_HIP(h, hipMalloc((void**)&h->chain_buf[i], need[i]));
        h->chain_bytes[i] = need[i];
        if (i >= 13) DS_HIP(h, hipMemset(h->chain_buf[i], 0, need[i]));       // delay_fbf starts from silence (SubbandGSC.py:111)
    }
    return DS_OK;
}

// launch the STFT of sub-handle `t` on dense channel-major input x [batch][C][n] -> Y [batch][T][K][C]
static int chain_stft(ds_handle* h, ds_handle* t, const float* x, int n, float* Y) {
    Params p;
    fill_params(t, p);
    const int C = t->cfg.n_mics, T = n / t->cfg.hop;
    p.x = x; p.y = Y;
    p.x_batch_stride = (long long)C * n; p.x_sample_stride = 1; p.x_chan_stride = n;
    p.y_batch_stride = (long long)T * t->K * C * 2;
    p.T = T; p.batch0 = 0;
    DS_HIP(h, t->ki.launch(p, t->cfg.batch, h->stream));
    return DS_OK;
}
static int chain_istft(ds_handle* h, ds_handle* t, const float* Y, int T, float* y, long long y_batch_stride) {
    Params p;
    fill_params(t, p);
    p.x = Y; p.y = y;
    p.x_batch_stride = (long long)T * t->K * 2;
    p.y_batch_stride = y_batch_stride;
    p.T = T; p.batch0 = 0; p.method = 1;
    DS_HIP(h, t->ki_istft.launch(p, t->cfg.batch, h->stream));
    return DS_OK;
}

// x_dev: [B][M][n] with element strides (x_bstride, x_cstride); y_dev [B] rows of n with stride y_bstride; the optional outputs dense
static int chain2_run(ds_handle* h, const float* x_dev, long long x_bstride, long long x_cstride, int n, float* y_dev, long long y_bstride,
                      float* fix_dev, float* bm_dev, float* p_dev, float* al_dev) {
    int rc = set_device(h); if (rc) return rc;
    ds_handle* fe = h->sub[0];
    const int B = h->cfg.batch, M = h->cfg.n_mics, K = h->K, hop = h->cfg.hop, T = n / hop;
    if (fe->aux_floats == 0 || fe->aux_floats % M != 0 || h->sub[2]->aux_floats < (size_t)K)
        return fail(h, DS_ESTATE, "SubbandGSC chain: call ds_chain_set_aux(DS_CHAIN_AUX_FIR) and (DS_CHAIN_AUX_COHERENCE) first");
    rc = chain2_reserve(h, n); if (rc) return rc;
    float** cb = h->chain_buf;
#define DS_SUB(i, call) do { int rc_ = (call); if (rc_) return fail(h, rc_, h->sub[i]->err); } while (0)
    {   // :177-178 DC notch per channel, then :201,206 TimeAlignment FIR bank + channel mean (the fixed beamformer)
        ds::TdParams p;
        std::memset(&p, 0, sizeof p);
        p.B = B; p.M = M; p.n = n; p.x = x_dev; p.x_bstride = x_bstride; p.x_cstride = x_cstride; p.y = cb[0]; p.mem = fe->td_mem;
        p.radius = fe->cfg.filt_alpha;
        DS_HIP(h, ds::launch_dcnotch(p, h->stream));
        const int Lt = (int)(fe->aux_floats / M);
        if (fe->td_L != Lt) {
            DS_HIP(h, hipStreamSynchronize(h->stream));
            for (int i = 0; i < 2; ++i) {
                (void)hipFree(fe->td_cache[i]); fe->td_cache[i] = nullptr;
                const size_t cbytes = (size_t)B * (Lt > 1 ? Lt - 1 : 1) * M * sizeof(float);
                DS_HIP(h, hipMalloc((void**)&fe->td_cache[i], cbytes));
                DS_HIP(h, hipMemset(fe->td_cache[i], 0, cbytes));
            }
            fe->td_L = Lt; fe->td_cur = 0;
        }
        std::memset(&p, 0, sizeof p);
        p.B = B; p.M = M; p.n = n; p.L = Lt; p.x = cb[0]; p.x_chan_major = 1; p.y = cb[1]; p.y_chan_major = 1; p.mean = cb[2];
        p.coef = fe->dev_buf[9]; p.cache_in = fe->td_cache[fe->td_cur]; p.cache_out = fe->td_cache[fe->td_cur ^ 1];
        DS_HIP(h, ds::launch_fir(p, h->stream));
        fe->td_cur ^= 1;
    }
    rc = chain_stft(h, h->sub[1], cb[1], n, cb[3]); if (rc) return rc;                                   // :204  D
    DS_SUB(2, ds_mcspp_estimate(h->sub[2], cb[3], T, cb[4], cb[5], nullptr, nullptr, nullptr, DS_MEM_DEVICE));   // :208  p
    rc = chain_stft(h, h->sub[3], cb[2], n, cb[6]); if (rc) return rc;                                   // bm[m].transform_x: F
    rc = chain_stft(h, h->sub[4], cb[1], n, cb[7]); if (rc) return rc;                                   // bm[m].transform_d analysis: B*M channels
    if (h->sub[5]->cfg.algo == DS_ALGO_SUBRLS) DS_SUB(5, ds_subrls_update(h->sub[5], cb[6], cb[7], T, cb[8], DS_MEM_DEVICE));
    else DS_SUB(5, ds_sublms_update(h->sub[5], cb[6], cb[7], cb[4], T, cb[8], DS_MEM_DEVICE));           // :217-223
    rc = chain_istft(h, h->sub[4], cb[8], T, cb[9], n); if (rc) return rc;                               // bm outputs, [B*M][n] = [B][M][n]
    rc = chain_stft(h, h->sub[6], cb[9], n, cb[10]); if (rc) return rc;                                  // :230-234  aic transform_x
    {   // :226 delay_fbf: the canceller's desired signal is the fixed output one block late = F shifted by one frame
        const size_t fr = (size_t)K * 8;
        if (T > 1) DS_HIP(h, hipMemcpy2DAsync((char*)cb[11] + fr, T * fr, cb[6], T * fr, (T - 1) * fr, B, hipMemcpyDeviceToDevice, h->stream));
        DS_HIP(h, hipMemcpy2DAsync(cb[11], T * fr, cb[13], fr, fr, B, hipMemcpyDeviceToDevice, h->stream));
        DS_HIP(h, hipMemcpy2DAsync(cb[13], fr, (char*)cb[6] + (T - 1) * fr, T * fr, fr, B, hipMemcpyDeviceToDevice, h->stream));
    }
    DS_SUB(7, ds_sublms_update(h->sub[7], cb[10], cb[11], cb[4], T, cb[12], DS_MEM_DEVICE));
    rc = chain_istft(h, h->sub[8], cb[12], T, y_dev, y_bstride); if (rc) return rc;
    {   // fix_output = fixed beamformer output delayed by one block (:226,255); the carried block is state either way
        const size_t blk = (size_t)hop * 4, row = (size_t)n * 4;
        if (fix_dev) {
            if (T > 1) DS_HIP(h, hipMemcpy2DAsync((char*)fix_dev + blk, row, cb[2], row, row - blk, B, hipMemcpyDeviceToDevice, h->stream));
            DS_HIP(h, hipMemcpy2DAsync(fix_dev, row, cb[14], blk, blk, B, hipMemcpyDeviceToDevice, h->stream));
        }
        DS_HIP(h, hipMemcpy2DAsync(cb[14], blk, (char*)cb[2] + (row - blk), row, blk, B, hipMemcpyDeviceToDevice, h->stream));
    }
    const size_t nb = (size_t)B * M * n * 4;
    if (bm_dev) DS_HIP(h, hipMemcpyAsync(bm_dev, cb[9], nb, hipMemcpyDeviceToDevice, h->stream));
    if (al_dev) DS_HIP(h, hipMemcpyAsync(al_dev, cb[1], nb, hipMemcpyDeviceToDevice, h->stream));
    if (p_dev) DS_HIP(h, hipMemcpyAsync(p_dev, cb[4], (size_t)B * T * K * 4, hipMemcpyDeviceToDevice, h->stream));
#undef DS_SUB
    return DS_OK;
}

int ds_chain_set_aux(ds_handle* h, int which, const float* table, size_t n_floats) {
    if (!h || !table) return fail(h, DS_EINVAL, "ds_chain_set_aux: NULL argument");
    if (h->cfg.algo != DS_ALGO_SUBBAND_GSC) return fail(h, DS_ESTATE, "ds_chain_set_aux: handle is not a DS_ALGO_SUBBAND_GSC object");
    ds_handle* t = which == DS_CHAIN_AUX_FIR ? h->sub[0] : which == DS_CHAIN_AUX_COHERENCE ? h->sub[2] : nullptr;
    if (!t) return fail(h, DS_EINVAL, "ds_chain_set_aux: unknown table id");
    const int rc = ds_set_aux(t, table, n_floats);
    return rc ? fail(h, rc, t->err) : DS_OK;
}

int ds_subband_gsc_process(ds_handle* h, const float* x, int n_samples, float* y, float* fix_output, float* bm_output, float* pp,
                           float* aligned, int mem) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_subband_gsc_process: NULL argument");
    if (h->cfg.algo != DS_ALGO_SUBBAND_GSC) return fail(h, DS_ESTATE, "ds_subband_gsc_process: handle is not a DS_ALGO_SUBBAND_GSC object");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_subband_gsc_process: n_samples must be a multiple of hop");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, n = n_samples, T = n / h->cfg.hop;
    IoSpec io = {{x, nullptr, nullptr}, {B * M * n * 4, 0, 0}, {y, fix_output, bm_output, pp, aligned},
                 {B * n * 4, fix_output ? B * n * 4 : 0, bm_output ? B * M * n * 4 : 0, pp ? B * T * h->K * 4 : 0, aligned ? B * M * n * 4 : 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    rc = chain2_run(h, din[0], (long long)(M * n), (long long)n, n_samples, dout[0], (long long)n, fix_output ? dout[1] : nullptr,
                    bm_output ? dout[2] : nullptr, pp ? dout[3] : nullptr, aligned ? dout[4] : nullptr);
    if (rc) return rc;
    return io_end(h, mem, io, dout);
}

int ds_process_pcm16(ds_handle* h, const int16_t* pcm, int n_total_channels, int first_channel, int n_samples, int16_t* out) {
    if (!h || !pcm || !out) return fail(h, DS_EINVAL, "ds_process_pcm16: NULL argument");
    if (h->cfg.algo > DS_ALGO_GSC) return fail(h, DS_ESTATE, "ds_process_pcm16: handle is a frame-level object");
    const int M = h->cfg.n_mics;
    if (first_channel < 0 || first_channel + M > n_total_channels) return fail(h, DS_ESHAPE, "ds_process_pcm16: microphone channels outside the frame");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_process_pcm16: n_samples must be a multiple of hop");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, L = n_samples;
    const size_t xe = B * L * M, ye = B * L, pe = B * L * n_total_channels;
    rc = stage_reserve(h, 0, pe * sizeof(int16_t)); if (rc) return rc;        // raw PCM in
    rc = stage_reserve(h, 1, xe * sizeof(float)); if (rc) return rc;          // float mics [B][L][M]
    rc = stage_reserve(h, 4, ye * sizeof(float)); if (rc) return rc;          // enhanced float
    rc = stage_reserve(h, 5, ye * sizeof(int16_t)); if (rc) return rc;        // enhanced PCM
    DS_HIP(h, hipMemcpyAsync(h->dev_buf[0], pcm, pe * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
    DS_HIP(h, ds::launch_pcm16_to_float((const short*)h->dev_buf[0], h->dev_buf[1], (long long)xe, n_total_channels, first_channel, M, h->stream));
    rc = ds_process_device(h, h->dev_buf[1], DS_LAYOUT_SAMPLES_CHANNELS, (long long)(L * M), 0, n_samples, h->dev_buf[4], (long long)L, 0,
                           h->cfg.batch, nullptr);
    if (rc) return rc;
    DS_HIP(h, ds::launch_float_to_pcm16(h->dev_buf[4], (short*)h->dev_buf[5], (long long)ye, h->stream));
    DS_HIP(h, hipMemcpyAsync(out, h->dev_buf[5], ye * sizeof(int16_t), hipMemcpyDeviceToHost, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_synchronize(ds_handle* h) {
    if (!h) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_timing_begin(ds_handle* h) {
    if (!h) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipEventRecord(h->ev0, h->stream));
    return DS_OK;
}

int ds_timing_end(ds_handle* h, float* elapsed_ms) {
    if (!h || !elapsed_ms) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipEventRecord(h->ev1, h->stream));
    DS_HIP(h, hipEventSynchronize(h->ev1));
    DS_HIP(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return DS_OK;
}

size_t ds_field_bytes(const ds_handle* h, int field) {
    if (!h) return 0;
    const size_t B = h->cfg.batch, K = h->K, M = h->cfg.n_mics;
    const bool ad = h->cfg.algo == DS_ALGO_ADAPTIVE, gsc = h->cfg.algo == DS_ALGO_GSC;
    switch (field) {
        case DS_FIELD_RVV: return ad ? B * K * M * M * 2 * sizeof(float) : 0;
        case DS_FIELD_RYY: return (ad && h->cfg.track_ryy) ? B * K * M * M * 2 * sizeof(float) : 0;
        case DS_FIELD_MCRA_S: case DS_FIELD_MCRA_SMIN: case DS_FIELD_MCRA_STMP: case DS_FIELD_MCRA_P:
        case DS_FIELD_MCRA_LAMBDA_D: return ad ? B * K * sizeof(float) : 0;
        case DS_FIELD_PHI_YY: case DS_FIELD_PHI_VV: return gsc ? B * K * M * M * sizeof(float) : 0;
        case DS_FIELD_G_AIC: return gsc ? B * K * (M - 1) * 2 * sizeof(float) : 0;
        case DS_FIELD_STFT_TAIL: return tail_in_bytes(h);
        case DS_FIELD_OLA_TAIL: return tail_out_bytes(h);
        case DS_FIELD_COUNTERS: return counters_bytes(h);
        case DS_FIELD_OP_STATE:
            if (h->tdf_w) return (size_t)h->cfg.batch * h->cfg.filter_len * sizeof(float);
            return opst_bytes(h);
        default: return 0;
    }
}

int ds_get_state(ds_handle* h, int field, void* dst, size_t bytes) {
    if (!h || !dst) return fail(h, DS_EINVAL, "ds_get_state: NULL argument");
    const size_t need = ds_field_bytes(h, field);
    if (need == 0) return fail(h, DS_EINVAL, "ds_get_state: field not available for this algo/config");
    if (bytes != need) return fail(h, DS_ESHAPE, "ds_get_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    if (field == DS_FIELD_STFT_TAIL) { DS_HIP(h, hipMemcpy(dst, h->tail_in, need, hipMemcpyDeviceToHost)); return DS_OK; }
    if (field == DS_FIELD_OLA_TAIL) { DS_HIP(h, hipMemcpy(dst, h->tail_out, need, hipMemcpyDeviceToHost)); return DS_OK; }
    if (field == DS_FIELD_COUNTERS) {
        DS_HIP(h, hipMemcpy(dst, h->counters, need, hipMemcpyDeviceToHost));
        if (h->op >= 0) {                                 // operator handles keep uniform counters on the host
            int* c = (int*)dst;
            for (int b = 0; b < h->cfg.batch; ++b) { c[4 * b] = h->op_frm; c[4 * b + 1] = h->op_ell; c[4 * b + 2] = h->op_frm; c[4 * b + 3] = h->op_first; }
        }
        return DS_OK;
    }
    if (field == DS_FIELD_OP_STATE) {
        DS_HIP(h, hipMemcpy(dst, h->tdf_w ? (const void*)h->tdf_w : (const void*)h->opst, need, hipMemcpyDeviceToHost));
        return DS_OK;
    }
    // per-bin fields: pull the raw planes and unpack on the host
    std::vector<float> raw(bins_bytes(h) / sizeof(float));
    DS_HIP(h, hipMemcpy(raw.data(), h->bins, bins_bytes(h), hipMemcpyDeviceToHost));
    const int B = h->cfg.batch, K = h->K, M = h->cfg.n_mics, KP = h->KP, NP = h->NP;
    auto at = [&](int b, int k, int f) -> float {   // float f of bin k of utterance b
        return raw[(((size_t)b * NP + f / 4) * KP + k) * 4 + (f % 4)];
    };
    float* out = (float*)dst;
    auto herm = [&](int d0, int o0) {
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < M; ++i)
                    for (int j = 0; j < M; ++j) {
                        float re, im;
                        if (i == j) { re = at(b, k, d0 + i); im = 0.0f; }
                        else if (i < j) { int q = ds::off_index(i, j, M); re = at(b, k, o0 + 2 * q); im = at(b, k, o0 + 2 * q + 1); }
                        else { int q = ds::off_index(j, i, M); re = at(b, k, o0 + 2 * q); im = -at(b, k, o0 + 2 * q + 1); }
                        float* o = out + ((((size_t)b * K + k) * M + i) * M + j) * 2;
                        o[0] = re; o[1] = im;
                    }
    };
    auto scalar = [&](int f) {
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k) out[(size_t)b * K + k] = at(b, k, f);
    };
    auto sym = [&](int s0) {
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < M; ++i)
                    for (int j = 0; j < M; ++j) {
                        int q = i <= j ? ds::sym_index(i, j, M) : ds::sym_index(j, i, M);
                        out[(((size_t)b * K + k) * M + i) * M + j] = at(b, k, s0 + q);
                    }
    };
    const int MCS = M * M;
    switch (field) {
        case DS_FIELD_RVV: herm(0, M); break;
        case DS_FIELD_RYY: herm(M * M + 5, M * M + 5 + M); break;
        case DS_FIELD_MCRA_S: scalar(MCS + 0); break;
        case DS_FIELD_MCRA_SMIN: scalar(MCS + 1); break;
        case DS_FIELD_MCRA_STMP: scalar(MCS + 2); break;
        case DS_FIELD_MCRA_P: scalar(MCS + 3); break;
        case DS_FIELD_MCRA_LAMBDA_D: scalar(MCS + 4); break;
        case DS_FIELD_PHI_YY: sym(0); break;
        case DS_FIELD_PHI_VV: sym(M * (M + 1) / 2); break;
        case DS_FIELD_G_AIC:
            for (int b = 0; b < B; ++b)
                for (int k = 0; k < K; ++k)
                    for (int i = 0; i < 2 * (M - 1); ++i)
                        out[((size_t)b * K + k) * 2 * (M - 1) + i] = at(b, k, M * (M + 1) + i);
            break;
        default: return fail(h, DS_EINVAL, "ds_get_state: unknown field");
    }
    return DS_OK;
}

static size_t own_state_bytes(const ds_handle* h) {
    return bins_bytes(h) + tail_in_bytes(h) + tail_out_bytes(h) + counters_bytes(h) + opst_bytes(h) + 4 * sizeof(int);
}
static size_t chain_hist_bytes(const ds_handle* h) {
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC) return (size_t)h->cfg.batch * (h->K * 8 + h->cfg.hop * 4);
    return h->cfg.algo == DS_ALGO_WPE_MVDR ? (size_t)h->cfg.batch * (h->wpe_delay > 0 ? h->wpe_delay : 1) * h->K * h->cfg.n_mics * 8 : 0;
}

size_t ds_state_bytes(const ds_handle* h) {
    if (!h) return 0;
    size_t n = own_state_bytes(h) + chain_hist_bytes(h);
    for (int i = 0; i < 10; ++i) if (h->sub[i]) n += ds_state_bytes(h->sub[i]);
    return n;
}

int ds_export_state(ds_handle* h, void* dst, size_t bytes) {
    if (!h || !dst) return fail(h, DS_EINVAL, "ds_export_state: NULL argument");
    if (bytes != ds_state_bytes(h)) return fail(h, DS_ESHAPE, "ds_export_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    char* d = (char*)dst;
    if (bins_bytes(h)) DS_HIP(h, hipMemcpy(d, h->bins, bins_bytes(h), hipMemcpyDeviceToHost));
    d += bins_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->tail_in, tail_in_bytes(h), hipMemcpyDeviceToHost)); d += tail_in_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->tail_out, tail_out_bytes(h), hipMemcpyDeviceToHost)); d += tail_out_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->counters, counters_bytes(h), hipMemcpyDeviceToHost)); d += counters_bytes(h);
    if (opst_bytes(h)) DS_HIP(h, hipMemcpy(d, h->opst, opst_bytes(h), hipMemcpyDeviceToHost));
    d += opst_bytes(h);
    const int uc[4] = {h->op_frm, h->op_ell, h->op_first, h->hist_cur};
    std::memcpy(d, uc, sizeof uc);
    d += sizeof uc;
    for (int i = 0; i < 10; ++i)
        if (h->sub[i]) {
            const size_t n = ds_state_bytes(h->sub[i]);
            rc = ds_export_state(h->sub[i], d, n); if (rc) return fail(h, rc, h->sub[i]->err);
            d += n;
        }
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC) {
        rc = chain2_reserve(h, h->cfg.hop); if (rc) return rc;
        const size_t n13 = (size_t)h->cfg.batch * h->K * 8, n14 = (size_t)h->cfg.batch * h->cfg.hop * 4;
        DS_HIP(h, hipMemcpy(d, h->chain_buf[13], n13, hipMemcpyDeviceToHost));
        DS_HIP(h, hipMemcpy(d + n13, h->chain_buf[14], n14, hipMemcpyDeviceToHost));
    } else if (chain_hist_bytes(h)) {
        rc = chain_reserve(h, 1); if (rc) return rc;
        DS_HIP(h, hipMemcpy(d, h->chain_buf[6], chain_hist_bytes(h), hipMemcpyDeviceToHost));
    }
    return DS_OK;
}

int ds_import_state(ds_handle* h, const void* src, size_t bytes) {
    if (!h || !src) return fail(h, DS_EINVAL, "ds_import_state: NULL argument");
    if (bytes != ds_state_bytes(h)) return fail(h, DS_ESHAPE, "ds_import_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    const char* s = (const char*)src;
    if (bins_bytes(h)) DS_HIP(h, hipMemcpy(h->bins, s, bins_bytes(h), hipMemcpyHostToDevice));
    s += bins_bytes(h);
    DS_HIP(h, hipMemcpy(h->tail_in, s, tail_in_bytes(h), hipMemcpyHostToDevice)); s += tail_in_bytes(h);
    DS_HIP(h, hipMemcpy(h->tail_out, s, tail_out_bytes(h), hipMemcpyHostToDevice)); s += tail_out_bytes(h);
    DS_HIP(h, hipMemcpy(h->counters, s, counters_bytes(h), hipMemcpyHostToDevice)); s += counters_bytes(h);
    if (opst_bytes(h)) DS_HIP(h, hipMemcpy(h->opst, s, opst_bytes(h), hipMemcpyHostToDevice));
    s += opst_bytes(h);
    int uc[4];
    std::memcpy(uc, s, sizeof uc);
    s += sizeof uc;
    h->op_frm = uc[0]; h->op_ell = uc[1]; h->op_first = uc[2]; h->hist_cur = uc[3];
    for (int i = 0; i < 10; ++i)
        if (h->sub[i]) {
            const size_t n = ds_state_bytes(h->sub[i]);
            rc = ds_import_state(h->sub[i], s, n); if (rc) return fail(h, rc, h->sub[i]->err);
            s += n;
        }
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC) {
        rc = chain2_reserve(h, h->cfg.hop); if (rc) return rc;
        const size_t n13 = (size_t)h->cfg.batch * h->K * 8, n14 = (size_t)h->cfg.batch * h->cfg.hop * 4;
        DS_HIP(h, hipMemcpy(h->chain_buf[13], s, n13, hipMemcpyHostToDevice));
        DS_HIP(h, hipMemcpy(h->chain_buf[14], s + n13, n14, hipMemcpyHostToDevice));
    } else if (chain_hist_bytes(h)) {
        rc = chain_reserve(h, 1); if (rc) return rc;
        DS_HIP(h, hipMemcpy(h->chain_buf[6], s, chain_hist_bytes(h), hipMemcpyHostToDevice));
    }
    return DS_OK;
}

}  // extern "C"
