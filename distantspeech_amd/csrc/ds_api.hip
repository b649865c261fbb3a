// ds_api.hip — handle management and the C-ABI of libdsenh.so (see include/dsenh.h).
// No CPU compute path lives here: every ds_process* call launches the gfx950 kernels.
#include <cstdlib>

#include "ds_handle.hpp"

using namespace dsi;

namespace dsi {

thread_local std::string g_err;

int fail(ds_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_err = msg;
    return code;
}


// StateLayout::ust(): NF KP floats per utterance, rounded up to a 128-byte line
static size_t bins_ust(const ds_handle* h) { return ((size_t)h->ki.NF * h->KP + 31) & ~(size_t)31; }
size_t bins_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * bins_ust(h) * sizeof(float); }
// Transform.previous_input / previous_output hold n_fft - hop samples per channel (transform.py:424-426): one hop, or three for hop = n_fft / 4
size_t tail_in_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * h->cfg.n_mics * (h->cfg.nfft - h->cfg.hop) * sizeof(float); }
size_t tail_out_bytes(const ds_handle* h) {
    const size_t ch = h->cfg.algo == DS_ALGO_TRANSFORM ? (size_t)h->cfg.n_mics : 1;   // Transform keeps one OLA tail per channel
    return (size_t)h->cfg.batch * ch * (h->cfg.nfft - h->cfg.hop) * sizeof(float);
}
size_t opst_bytes(const ds_handle* h) { return h->op >= 0 ? (size_t)h->cfg.batch * op_ust(h) * sizeof(float) : 0; }
size_t counters_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * 4 * sizeof(int); }

int set_device(ds_handle* h) {
    DS_HIP(h, hipSetDevice(h->device));
    // every entry point comes through here before it touches the handle; the utterance groups of a chain are brought back onto the
    // chain's stream first (the chain's own launch path selects the device without this: select_device)
    if (h->groups_open && !h->group_enqueue) return join_groups(h);
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
        DS_HIP(h, hipMemsetAsync(h->td_mem, 0, (size_t)h->cfg.batch * h->cfg.n_mics * 2 * sizeof(double), h->stream));
        for (int i = 0; i < 2; ++i)
            if (h->td_cache[i]) DS_HIP(h, hipMemsetAsync(h->td_cache[i], 0, (size_t)h->cfg.batch * (h->td_L > 1 ? h->td_L - 1 : 1) * h->cfg.n_mics * sizeof(float), h->stream));
    }
    if (h->op == ds::OP_WPE && h->NF > 0) {
        // awpe.py:58-77: P = I * 1e-3, everything else zero — on the device (the state of a wide filter is gigabytes: 3.8 MB per utterance
        // at 4 channels x 20 taps, 129 bins)
        DS_HIP(h, ds::launch_wpe_init(h->opst, h->cfg.batch, h->K, (long long)op_ust(h), h->cfg.n_mics, h->filter_len, h->stream));
        if (h->wpe64) DS_HIP(h, ds::launch_wpe64_init(h->wpe64, h->cfg.batch, h->K, (long long)wpe64_ust(h), h->cfg.n_mics, h->filter_len, h->stream));
        h->op_frm = 0; h->op_ell = 1; h->op_first = 1; h->wpe_started = false;
    } else if (h->op >= 0 && h->NF > 0) {
        // operator state: zeros, except the rows the reference initialises to non-zero values
        std::vector<float> st((size_t)h->cfg.batch * op_ust(h), 0.0f);
        auto fill_row = [&](int f, float v) {                // row f of every bin (float4 planes: ds_ops.hpp st_index)
            for (int b = 0; b < h->cfg.batch; ++b)
                for (int k = 0; k < h->KP; ++k) st[(size_t)ds::st_index(b, f, k, h->NF, h->KP)] = v;
        };
        if (h->op == ds::OP_OMLSA) {                       // omlsa_multi.py:33-58: gamma, G_H1, G, xi_hat, q_hat = 1
            const int o_s = 5 * h->cfg.n_mics + 1 + (h->cfg.n_mics - 1);
            fill_row(o_s + 1, 1.0f); fill_row(o_s + 2, 1.0f); fill_row(o_s + 3, 1.0f); fill_row(o_s + 5, 1.0f); fill_row(o_s + 6, 1.0f);
            fill_row(5 * h->cfg.n_mics, 1.0f);             // zeta_Y = 1
        }
        if (h->op == ds::OP_SUBRLS) {                      // SubbandRLS.py:40-42: P = I / 1e-3
            const int N = h->filter_len;
            for (int i = 0; i < N; ++i) fill_row(4 * N + 2 * (i * N + i), 1000.0f);
        }
        DS_HIP(h, hipMemcpyAsync(h->opst, st.data(), st.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        h->op_frm = 0; h->op_ell = 1; h->op_first = 1;
    }
    h->td_cur = 0;
    { const int rc = sync_dev_cnt(h); if (rc) return rc; }
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

// host mirror of the uniform counters (and the aux word: FIR parity / WPE ring position) -> the device copy the kernels read
int sync_dev_cnt(ds_handle* h) {
    if (!h->dev_cnt) return DS_OK;
    const int aux = h->cfg.algo == DS_ALGO_FRONTEND ? h->td_cur : wpe_chain(h) ? h->hist_cur : 0;
    int c[8 * DS_GROUPS];                                          // one copy per utterance group of a chain (groups run at their own pace)
    for (int g = 0; g < DS_GROUPS; ++g) { int* q = c + 8 * g; q[0] = h->op_frm; q[1] = h->op_ell; q[2] = h->op_first; q[3] = aux; q[4] = q[5] = q[6] = q[7] = 0; }
    DS_HIP(h, hipMemcpyAsync(h->dev_cnt, c, sizeof c, hipMemcpyHostToDevice, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));                      // c is a stack buffer
    return DS_OK;
}

int post_tick(ds_handle* t, int* cnt, int frames, int L, int aux_add, int aux_mod, hipStream_t stream) {
    ds_handle* o = t->owner ? t->owner : t;
    if (o->pend_set) { const int rc = flush_tick(o); if (rc) return rc; }
    o->pend = ds::TickArgs{cnt, frames, L > 0 ? L : 1, aux_add, aux_mod};
    o->pend_stream = stream; o->pend_set = true;
    if (!t->owner) return flush_tick(o);                             // a stand-alone stage has no next launch to count on
    return DS_OK;
}
void take_tick(ds_handle* t, hipStream_t stream, ds::TickArgs& out) {
    ds_handle* o = t->owner ? t->owner : t;
    out = ds::TickArgs{nullptr, 0, 1, 0, 0};
    if (o->pend_set && o->pend_stream == stream) { out = o->pend; o->pend_set = false; }
}
int flush_tick(ds_handle* o) {
    if (!o->pend_set) return DS_OK;
    o->pend_set = false;
    DS_HIP(o, ds::launch_tick(o->pend.cnt, o->pend.frames, o->pend.L, o->pend.aux_add, o->pend.aux_mod, o->pend_stream));
    return DS_OK;
}

// utterance groups of a chain run on their own streams between calls; anything else that touches the handle first orders the chain's
// stream behind them
int join_groups(ds_handle* h) {
    if (!h || !h->groups_open) return DS_OK;
    h->groups_open = false;
    for (int g = 0; g < 7; ++g) {
        if (!h->side[g] || !h->ev_join[g]) continue;
        DS_HIP(h, hipEventRecord(h->ev_join[g], h->side[g]));
        DS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join[g], 0));
    }
    return DS_OK;
}

void advance_host_counters(ds_handle* h, int frames, int L) {
    for (int t = 0; t < frames; ++t) {
        if (h->op_frm != 0 && h->op_ell % L == 0) h->op_ell = 0;
        h->op_frm += 1; h->op_ell += 1;
    }
    if (frames > 0) h->op_first = 0;
}

hipError_t launch_transform_stft(const ds_handle* t, const Params& p, int batch, hipStream_t stream) {
    if (t->cfg.n_mics == 1 && t->ki_rows.launch && p.batch0 == 0) return t->ki_rows.launch(p, batch, stream);
    return t->ki.launch(p, batch, stream);
}
hipError_t launch_transform_istft(const ds_handle* t, const Params& p, int batch, hipStream_t stream) {
    if (t->cfg.n_mics == 1 && t->ki_rows_istft.launch && p.batch0 == 0 && p.method == 1) return t->ki_rows_istft.launch(p, batch, stream);
    return t->ki_istft.launch(p, batch, stream);
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
    p.beta_y = ds::complement_of(h->alpha_y);
    p.beta_v = ds::complement_of(h->alpha_v);
    p.diag = h->diag; p.diag_floor = ds::pivot_floor(h->diag);
    p.gate = h->gate;
    p.mu = h->mu;
}

// host mirrors of the uniform counters of a chain handle (index 10) and its stages (0..9)
struct ChainMirrors { int frm[11], ell[11], first[11], td[11], hist; };
static void mirrors_get(const ds_handle* h, ChainMirrors& m) {
    for (int i = 0; i < 11; ++i) {
        const ds_handle* t = i < 10 ? h->sub[i] : h;
        m.frm[i] = t ? t->op_frm : 0; m.ell[i] = t ? t->op_ell : 0; m.first[i] = t ? t->op_first : 0; m.td[i] = t ? t->td_cur : 0;
    }
    m.hist = h->hist_cur;
}
static void mirrors_set(ds_handle* h, const ChainMirrors& m) {
    for (int i = 0; i < 11; ++i) {
        ds_handle* t = i < 10 ? h->sub[i] : h;
        if (!t) continue;
        t->op_frm = m.frm[i]; t->op_ell = m.ell[i]; t->op_first = m.first[i]; t->td_cur = m.td[i];
    }
    h->hist_cur = m.hist;
}
// forget every cached hipGraph of a handle (a captured sequence holds device addresses and kernel choices of the moment it was captured)
static void drop_graphs(ds_handle* h) {
    h->graph_valid = false; h->chain_warm_n = -1;
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    for (int g = 0; g < 8; ++g) if (h->group_exec[g]) { (void)hipGraphExecDestroy(h->group_exec[g]); h->group_exec[g] = nullptr; }
}
static void advance_host_counters_keep_first(ds_handle* h, int frames, int L) {
    const int first = h->op_first;
    advance_host_counters(h, frames, L);
    if (h->cfg.algo == DS_ALGO_MCSPP) h->op_first = first;           // ds_mcspp_estimate leaves it alone
}

}  // namespace dsi

#ifndef DS_ARCH
#define DS_ARCH gfx950
#endif
#define DS_STR2(x) #x
#define DS_STR(x) DS_STR2(x)

extern "C" {

int ds_version(void) { return DS_VERSION; }

const char* ds_build_info(void) {
#if defined(DS_WITH_SHELVED)
    return "libdsenh version=" DS_STR(DS_VERSION) " state_layout=" DS_STR(DS_STATE_LAYOUT) " arch=" DS_STR(DS_ARCH) " shelved=1";
#else
    return "libdsenh version=" DS_STR(DS_VERSION) " state_layout=" DS_STR(DS_STATE_LAYOUT) " arch=" DS_STR(DS_ARCH) " shelved=0";
#endif
}

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

// DS_CHAIN_PRIO=<mask> (A/B runs; default 0 = none): which streams of a SubbandGSC chain run at the device's greatest priority — 1 the chain's own
// (McSpp), 2 the tail's, 4 the front end's, 8 the blocking filters'.  The tail's stream first was +2.0 .. +3.7 % at one block per call in eleven
// interleaved pairs on three boxes and -4 .. -7 % in the second half of a fourth session: inside the run-to-run spread of the chain itself (a run lands
// on one of two levels, 8.3 - 8.7 or 9.3 - 9.6 M frames/s, with every kernel of the step scaled alike: the device's clock state,
// profiles/r03f/cfg5_step_length.txt) — not the default (profiles/r03f/chain_prio_ab.txt)
static hipError_t chain_stream(hipStream_t* s, int bit) {
#ifdef DS_ABLATE_CHAIN   // timing experiment only: the chain's streams on disjoint sets of CUs (DS_ABL_CU_<bit>="lo-hi": mask bits lo .. hi-1)
    { char name[32]; std::snprintf(name, sizeof name, "DS_ABL_CU_%d", bit);
      if (const char* e = std::getenv(name)) {
          int lo = 0, hi = 0;
          if (std::sscanf(e, "%d-%d", &lo, &hi) == 2 && lo >= 0 && hi > lo && hi <= 256) {
              uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
              for (int i = lo; i < hi; ++i) mask[i >> 5] |= 1u << (i & 31);
              return hipExtStreamCreateWithCUMask(s, 8, mask);
          }
      } }
#endif
    const char* pr = std::getenv("DS_CHAIN_PRIO");
    const int mask = pr ? std::atoi(pr) : 0;
    int lo = 0, hi = 0;
    if ((mask & bit) && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

int ds_create(const ds_config* cfg, ds_handle** out) {
    if (!cfg || !out) return fail(nullptr, DS_EINVAL, "ds_create: NULL argument");
    if (cfg->struct_size != (int32_t)sizeof(ds_config)) return fail(nullptr, DS_EINVAL, "ds_create: struct_size mismatch");
    *out = nullptr;
    if (cfg->batch <= 0) return fail(nullptr, DS_EINVAL, "ds_create: batch must be > 0");
    if (dsi::frames_algo(cfg->algo) && cfg->hop * 2 != cfg->nfft)
        return fail(nullptr, DS_EUNSUPPORTED, "ds_create: the beamformer objects take hop == nfft/2 (the only overlap their reference callers use)");
    if (cfg->algo == DS_ALGO_TRANSFORM && cfg->hop * 2 != cfg->nfft && cfg->hop * 4 != cfg->nfft)
        return fail(nullptr, DS_EUNSUPPORTED, "ds_create: Transform takes hop == nfft/2 or hop == nfft/4");
    if (cfg->algo == DS_ALGO_ADAPTIVE_PF && cfg->track_ryy)
        return fail(nullptr, DS_EUNSUPPORTED, "ds_create: DS_ALGO_ADAPTIVE_PF keeps no Ryy (track_ryy must be 0; TFGSC is not one of its methods)");
    KernelInfo ki = {nullptr, 0, 0, 0};
    KernelInfo ki_istft = {nullptr, 0, 0, 0};
    int op = -1, NF = 0;
    const int KPo = ds::plane_len(cfg->nfft / 2 + 1);
    const int flen = cfg->filter_len > 0 ? cfg->filter_len : 2;
    const bool is_tdf = cfg->algo == DS_ALGO_TDNLMS || cfg->algo == DS_ALGO_TDRLS;
    switch (cfg->algo) {
        case DS_ALGO_FIXED: ki = ds::lookup_fixed(cfg->nfft, cfg->n_mics); break;
        case DS_ALGO_ADAPTIVE:
            ki = cfg->track_ryy ? ds::lookup_adaptive_ryy(cfg->nfft, cfg->n_mics) : ds::lookup_adaptive_noryy(cfg->nfft, cfg->n_mics);
#if defined(DS_WITH_SHELVED)
            if (!cfg->track_ryy) {                          // 8 microphones, DS_M8_QUAD=1: the quad-spread kernel (same state layout, same numbers)
                const KernelInfo kq = ds::lookup_adaptive_quad(cfg->nfft, cfg->n_mics);
                if (kq.launch) ki = kq;
            }
#endif
            break;
        case DS_ALGO_GSC: ki = ds::lookup_gsc(cfg->nfft, cfg->n_mics); break;
        case DS_ALGO_ADAPTIVE_PF: ki = ds::lookup_adaptive_pf(cfg->nfft, cfg->n_mics); ki.launch_long = ds::lookup_adaptive_pf_long(cfg->nfft, cfg->n_mics); break;
        case DS_ALGO_TRANSFORM:
            ki = ds::lookup_stft(cfg->nfft, cfg->n_mics, cfg->nfft / cfg->hop);
            ki_istft = ds::lookup_istft(cfg->nfft, cfg->n_mics, cfg->nfft / cfg->hop);
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
            if (ds::op_supported(ds::OP_MCSPP, cfg->n_mics) && cfg->n_mics >= 4 && cfg->n_mics <= 6 && cfg->hop * 2 == cfg->nfft && flen <= ds::RLS_NMAX) { op = 105; NF = 0; }
            break;
        case DS_ALGO_TDGSC: case DS_ALGO_FDGSC:
            if (cfg->n_mics >= 2 && cfg->n_mics <= 8 && cfg->hop * 2 == cfg->nfft &&
                (cfg->nfft == 128 || cfg->nfft == 256 || cfg->nfft == 512 || cfg->nfft == 1024)) { op = 106; NF = 0; }
            break;
        case DS_ALGO_MCSPP_MVDR:
            if (ds::op_supported(ds::OP_MCSPP, cfg->n_mics) && cfg->n_mics >= 3 && cfg->hop * 2 == cfg->nfft &&
                ds::lookup_stft(cfg->nfft, cfg->n_mics, 2).launch) { op = 108; NF = 0; }
            break;
        case DS_ALGO_WPE_MVDR:
            if (ds::op_supported(ds::OP_ADAPTIVE, cfg->n_mics) && cfg->hop * 2 == cfg->nfft && cfg->n_mics * flen <= ds::WPEW_CNMAX) { op = 104; NF = 0; }
            break;
        case DS_ALGO_WPE_TD:
            if (cfg->n_mics >= 1 && cfg->n_mics <= ds::WPE_CMAX && cfg->n_mics * flen <= ds::WPEW_CNMAX &&
                (cfg->hop * 2 == cfg->nfft || cfg->hop * 4 == cfg->nfft)) { op = 107; NF = 0; }
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
            if (cfg->n_mics >= 1 && cfg->n_mics <= ds::WPE_CMAX && cfg->n_mics * flen <= ds::WPEW_CNMAX) {
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
        snprintf(buf, sizeof buf, "ds_create: no kernel for algo=%d nfft=%d n_mics=%d (nfft in {256,512,1024}, n_mics in {2,4,6,8}; Transform: n_mics 1..8)",
                 cfg->algo, cfg->nfft, cfg->n_mics);
        return fail(nullptr, DS_EUNSUPPORTED, buf);
    }
    if (cfg->algo >= DS_ALGO_MCRA && !dsi::frames_algo(cfg->algo) && op < 0) return fail(nullptr, DS_EUNSUPPORTED, "ds_create: unsupported n_mics / filter_len for this frame-level object");
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
    h->graph_exec = nullptr; h->graph_valid = false; h->chain_warm_n = -1; h->adv_hist = 0;
    for (int i = 0; i < 8; ++i) h->group_exec[i] = nullptr;
    for (int i = 0; i < 11; ++i) { h->adv_frames[i] = 0; h->adv_td[i] = 0; }
    // fused frame kernels: two free-running utterance groups from 2048 utterances up (more than one round of workgroups per launch)
    h->split = (dsi::frames_algo(cfg->algo) && cfg->batch >= 2048) ? 2 : 1; h->ev_fork = nullptr;
    h->parts = 1; h->groups_open = false;
    h->front_async = false; h->tail_async = false; h->front_open = false; h->fr_valid[0] = h->fr_valid[1] = false; h->tf_valid[0] = h->tf_valid[1] = false; h->bf_valid[0] = h->bf_valid[1] = false; h->al_read[0] = h->al_read[1] = false; h->fr_mid[0] = h->fr_mid[1] = false; h->front_set = 0; h->lean_main = false; h->early_front = false; h->fan_fused = false;
    for (int i = 0; i < 10; ++i) h->ev_fr[i] = nullptr;
    for (int i = 0; i < 7; ++i) { h->side[i] = nullptr; h->ev_join[i] = nullptr; }
    h->ki_istft = ki_istft; h->op = op;
    h->ki_rows = KernelInfo{nullptr, 0, 0, 0}; h->ki_rows_istft = h->ki_rows; h->ki_aic = h->ki_rows; h->ki_cdr = h->ki_rows; h->front_fused = false;
    if (cfg->algo == DS_ALGO_TRANSFORM && cfg->n_mics == 1 && cfg->hop * 2 == cfg->nfft) { h->ki_rows = ds::lookup_stft_rows(cfg->nfft); h->ki_rows_istft = ds::lookup_istft_rows(cfg->nfft); }
 h->opst = nullptr; h->NF = NF;
    h->op_frm = 0; h->op_ell = 1; h->op_first = 1; h->wpe_started = false;
    h->filter_len = flen; h->norm = cfg->no_norm ? 0 : 1;
    h->filt_mu = cfg->filt_mu > 0 ? cfg->filt_mu : (cfg->algo == DS_ALGO_SUBRLS ? 0.5f : 0.1f);
    h->filt_alpha = cfg->filt_alpha > 0 ? cfg->filt_alpha : 0.9f;
    h->rls_lambda = cfg->rls_lambda > 0 ? cfg->rls_lambda : 0.998f;
    for (int i = 0; i < 10; ++i) { h->dev_buf[i] = nullptr; h->dev_buf_bytes[i] = 0; }
    h->aux_floats = 0;
    h->tdf_w = h->tdf_buf = h->tdf_P = nullptr;
    h->mcspp_repeat = 0;
    h->dev_cnt = nullptr; h->use_dev_cnt = false;
    h->owner = nullptr; h->pend_set = false; h->pend_stream = nullptr; h->pend = ds::TickArgs{nullptr, 0, 1, 0, 0};
    h->x_fan = 1; h->p_complement = 0; h->d_interleaved = 0; h->d_prev = nullptr;
    h->fdaf_kind = DS_FDAF_PLAIN; h->fdaf_constrain = 1; h->fdaf_non_causal = 0; h->fdaf_weight_norm = 0; h->fdaf_two_path = 0;
    for (int i = 0; i < 10; ++i) h->sub[i] = nullptr;
    for (int i = 0; i < 24; ++i) { h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0; }
    h->postfilter = 0;
    h->owns_stream = true; h->wpe_delay = 4; h->hist_cur = 0; h->group_enqueue = false;
    h->wpe_only = cfg->algo == DS_ALGO_WPE_TD; h->wpe64 = nullptr;
    { const char* e = getenv("DS_WPE_GENERIC"); h->wpe_generic = (e && e[0] == '1') ? 1 : 0; }
    h->td_mem = nullptr; h->td_cache[0] = h->td_cache[1] = nullptr; h->td_L = 0; h->td_cur = 0;
    h->method = DS_METHOD_MVDR;
    h->mcra_L = cfg->mcra_L > 0 ? cfg->mcra_L : 15;
    {   // DS_PIPE_MIN_T=<hops>: calls of at least that many hops take the hop-pipelined frame kernel (ds_pipe.hpp).  Off by default: it is
        // bit-identical and measured SLOWER (cfg2, 625 hops per call: 193 against 216 M frames/s; profiles/r03b/pipe_ab.txt and
        // DESIGN.md section 4.4) — kept as an opt-in experiment with its tests
        const char* e = getenv("DS_PIPE_MIN_T");
        const int v = e ? atoi(e) : 0;
        h->pipe_min_T = v > 0 ? v : 0x7fffffff;
    }
    h->alpha_y = cfg->alpha_y > 0 ? cfg->alpha_y : 0.8f;
    h->alpha_v = cfg->alpha_v > 0 ? cfg->alpha_v : 0.9998f;
    h->diag = cfg->diag > 0 ? cfg->diag : 1e-6f;
    h->gate = cfg->gate > 0 ? cfg->gate : 0.4f;
    h->mu = cfg->mu > 0 ? cfg->mu : 0.01f;
    h->est_pos = -1; h->est_used = 0;
    h->ref_powers = false; h->ref_pow = nullptr; h->ref_pow_cap = 0; h->ref_pow_T = 0; h->ref_pow_stream = nullptr;
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
    DS_CRE(cfg->algo == DS_ALGO_SUBBAND_GSC ? chain_stream(&h->stream, 1) : hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    DS_CRE(hipEventCreate(&h->ev0));
    DS_CRE(hipEventCreate(&h->ev1));
    if (bins_bytes(h)) DS_CRE(hipMalloc((void**)&h->bins, bins_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->tail_in, tail_in_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->tail_out, tail_out_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->counters, counters_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->dev_cnt, 8 * DS_GROUPS * sizeof(int)));
    if (is_tdf) {
        const size_t Lf = cfg->filter_len;
        DS_CRE(hipMalloc((void**)&h->tdf_w, (size_t)cfg->batch * Lf * sizeof(float)));
        DS_CRE(hipMalloc((void**)&h->tdf_buf, (size_t)cfg->batch * Lf * sizeof(float)));
        if (cfg->algo == DS_ALGO_TDRLS) DS_CRE(hipMalloc((void**)&h->tdf_P, (size_t)cfg->batch * Lf * Lf * sizeof(float)));
        h->filt_mu = cfg->filt_mu > 0 ? cfg->filt_mu : (cfg->algo == DS_ALGO_TDRLS ? 0.5f : 0.1f);
        h->rls_lambda = cfg->rls_lambda > 0 ? cfg->rls_lambda : 0.9998f;                 // RLS.py:15
    }
    if (cfg->algo == DS_ALGO_FRONTEND) {
        DS_CRE(hipMalloc((void**)&h->td_mem, (size_t)cfg->batch * cfg->n_mics * 2 * sizeof(double)));
        DS_CRE(hipMemset(h->td_mem, 0, (size_t)cfg->batch * cfg->n_mics * 2 * sizeof(double)));
    }
    if (h->op >= 0 && h->NF > 0) DS_CRE(hipMalloc((void**)&h->opst, (size_t)cfg->batch * op_ust(h) * sizeof(float)));
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
    if (cfg->algo == DS_ALGO_WPE_MVDR || cfg->algo == DS_ALGO_WPE_TD) {
        const int algos[5] = {DS_ALGO_TRANSFORM, DS_ALGO_WPE, DS_ALGO_MCMCRA, DS_ALGO_ADAPTIVE_FRAMES, DS_ALGO_TRANSFORM};
        for (int i = 0; i < 5; ++i) {
            if (h->wpe_only && (i == 2 || i == 3)) continue;            // Wpe.update alone: no speech-presence / beamformer stages
            ds_config c = *cfg;
            c.algo = algos[i]; c.device = h->device;
            if (i == 4) c.n_mics = 1;
            rc = ds_create(&c, &h->sub[i]);
            if (rc != DS_OK) { std::string m = g_err; ds_destroy(h); return fail(nullptr, rc, "ds_create(WPE chain): stage " + std::to_string(i) + ": " + m); }
            (void)hipStreamDestroy(h->sub[i]->stream);          // every stage runs on the chain's stream
            h->sub[i]->stream = h->stream; h->sub[i]->owns_stream = false;
            h->sub[i]->use_dev_cnt = true;                      // uniform counters read from the device: the chain replays as a hipGraph
            h->sub[i]->owner = h;
        }
        // utterance groups: the batch as `parts` independent chains on their own streams (DS_PARAM_SPLIT; DS_CHAIN_PARTS in the environment
        // overrides the default for A/B runs).  Two groups from 512 utterances up: one group's HBM-bound WPE kernel runs next to the other
        // group's analysis, McMcra, MVDR and synthesis stages (+10 % at 1024 utterances); more groups cost the WPE kernel its grid
        h->parts = cfg->batch >= 512 ? 2 : 1;
        if (const char* e = std::getenv("DS_CHAIN_PARTS")) { const int v = std::atoi(e); if (v >= 1 && v <= 8) h->parts = v; }
    }
    if (cfg->algo == DS_ALGO_MCSPP_MVDR) {
        // mvdr.ipynb cell 4: transform = Transform(n_fft, hop, channel=M); noise_estimator = McSpp(nfft, channels=M); the same transform's istft
        const int algos[3] = {DS_ALGO_TRANSFORM, DS_ALGO_MCSPP, DS_ALGO_TRANSFORM};
        for (int i = 0; i < 3; ++i) {
            ds_config c = *cfg;
            c.algo = algos[i]; c.device = h->device;
            if (i == 2) c.n_mics = 1;
            rc = ds_create(&c, &h->sub[i]);
            if (rc != DS_OK) { std::string m = g_err; ds_destroy(h); return fail(nullptr, rc, "ds_create(DS_ALGO_MCSPP_MVDR): stage " + std::to_string(i) + ": " + m); }
            (void)hipStreamDestroy(h->sub[i]->stream);
            h->sub[i]->stream = h->stream; h->sub[i]->owns_stream = false;
            h->sub[i]->use_dev_cnt = true;                      // uniform counters on the device: a sequence of calls replays as a hipGraph
            h->sub[i]->owner = h;
        }
        const char* unf = std::getenv("DS_CHAIN_UNFUSED");
        if (!(unf && unf[0] == '1')) h->ki_cdr = ds::lookup_stft_cdr(cfg->nfft, cfg->n_mics);
    }
    if (cfg->algo == DS_ALGO_TDGSC || cfg->algo == DS_ALGO_FDGSC) {
        rc = gsc_chain_create(h);
        if (rc != DS_OK) { std::string m = h->err.empty() ? g_err : h->err; ds_destroy(h); return fail(nullptr, rc, m); }
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
            h->sub[i]->use_dev_cnt = true;
            h->sub[i]->owner = h;
        }
        const char* no_fork = std::getenv("DS_CHAIN_NO_FORK");               // A/B switch: every stage on the chain's own stream
        if (rls && !(no_fork && no_fork[0] == '1')) {
            // the RLS blocking filters do not take the speech presence probability, so the blocking-filter stages (HBM-bound) run on a side
            // stream next to the McSpp stage (register-bound, one wave per SIMD) and join in front of the canceller.  McSpp is the longer
            // branch and stays on the chain's own stream: the cross-stream hand-offs (~10 us each) then sit on the branch that has slack
            if (chain_stream(&h->side[0], 8) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_join[0], hipEventDisableTiming) != hipSuccess) {
                ds_destroy(h);
                return fail(nullptr, DS_EHIP, "ds_create(DS_ALGO_SUBBAND_GSC): side stream");
            }
            for (int i = 3; i <= 6; ++i) h->sub[i]->stream = h->side[0];
        }
        // the front end (notch, FIR bank, M-channel analysis: latency-bound) on a stream of its own, double-buffered: block t + 1's front end
        // next to block t's HBM-bound stages (DS_CHAIN_SERIAL_FRONT=1: on the chain's stream, A/B runs)
        const char* serial = std::getenv("DS_CHAIN_SERIAL_FRONT");
        if (!(serial && serial[0] == '1')) {
            bool ok = chain_stream(&h->side[1], 4) == hipSuccess;
            for (int i = 0; i < 10 && ok; ++i) ok = hipEventCreateWithFlags(&h->ev_fr[i], hipEventDisableTiming) == hipSuccess;
            // the tail on a stream of its own (side[2]) pays only when that stream gets a hardware queue of its own: with the HIP runtime's
            // default of 4 queues per device the fourth and fifth stream of the process share one and the front end queues behind the
            // previous block's tail (measured: 0.31 ms per block instead of 0.28).  The library cannot see how many queues the running
            // runtime has, so the APPLICATION says so: DS_PARAM_TAIL_ASYNC = 1 before the first call (or DS_CHAIN_TAIL_ASYNC=1 in the
            // environment at ds_create) after it has raised the limit itself before its first HIP call (GPU_MAX_HW_QUEUES=8; bench.py
            // does).  Default 0: the tail stays on the chain's stream
            if (ok) ok = chain_stream(&h->side[2], 2) == hipSuccess && hipEventCreateWithFlags(&h->ev_join[2], hipEventDisableTiming) == hipSuccess;
            const char* ta = std::getenv("DS_CHAIN_TAIL_ASYNC");
            h->tail_async = ok && ta && ta[0] == '1';
            if (ok && !h->ev_fork) ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
            if (!ok) { ds_destroy(h); return fail(nullptr, DS_EHIP, "ds_create(DS_ALGO_SUBBAND_GSC): front-end stream"); }
            h->front_async = true;
            h->sub[0]->stream = h->side[1]; h->sub[1]->stream = h->side[1];
            // a pipelined chain is never replayed as a graph, so McSpp takes its frame counters by value (the host mirror): no
            // counter-advance launch between two McSpp launches, which are the chain's critical loop (DS_CHAIN_MAIN_JOIN=1: the older
            // arrangement, A/B runs)
            const char* mj = std::getenv("DS_CHAIN_MAIN_JOIN");
            h->lean_main = !(mj && mj[0] == '1');
            if (h->lean_main) h->sub[2]->use_dev_cnt = false;
            const char* ne = std::getenv("DS_CHAIN_NO_EARLY");
            h->early_front = !(ne && ne[0] == '1');
            // ... with McCDR as the per-bin program of the analysis kernel (the workgroup holds the utterance: MCRA stencil and band mean from
            // LDS); counters by value, so only where nothing is replayed as a graph.  DS_CHAIN_UNFUSED=1: the separate McCDR launch
            const char* unf2 = std::getenv("DS_CHAIN_UNFUSED");
            if (!(unf2 && unf2[0] == '1')) h->ki_cdr = ds::lookup_stft_cdr(cfg->nfft, M);
            const char* ff = std::getenv("DS_CHAIN_FRONT_FUSED");        // shelved (make SHELVED=1): the front end as ONE kernel, measured slower
            h->front_fused = h->ki_cdr.launch != nullptr && ff && ff[0] == '1';
        }
#if defined(DS_WITH_SHELVED)
        { const char* fs = std::getenv("DS_CHAIN_FAN_FUSED"); h->fan_fused = fs && fs[0] == '1'; }
#endif
        h->sub[5]->x_fan = M;                 // the M blocking filters of an utterance share its fixed-beamformer spectrum and p ...
        h->sub[5]->d_interleaved = 1;         // ... and take their desired signals straight from the M-channel STFT of the aligned channels
        h->sub[7]->p_complement = 1;          // SubbandGSC.py:232: p = 1 - p
        // the chain's tail (re-analysis of the blocking-matrix outputs -> canceller -> synthesis) as one frame kernel on the stages' own
        // state; DS_CHAIN_UNFUSED=1 keeps the three separate kernels (A/B runs, and the reference point of the parity test)
        const char* unf = std::getenv("DS_CHAIN_UNFUSED");
        if (flen == 2 && h->sub[7]->NF == 8 * M + 1 && !(unf && unf[0] == '1')) h->ki_aic = ds::lookup_aic(cfg->nfft, M);
    }
    *out = h;
    return DS_OK;
}

int ds_destroy(ds_handle* h) {
    if (!h) return DS_EINVAL;
    (void)hipSetDevice(h->device);
    for (int i = 0; i < 7; ++i) if (h->side[i]) (void)hipStreamSynchronize(h->side[i]);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < 10; ++i) if (h->sub[i]) (void)ds_destroy(h->sub[i]);
    for (int i = 0; i < 24; ++i) (void)hipFree(h->chain_buf[i]);
    (void)hipFree(h->bins); (void)hipFree(h->tail_in); (void)hipFree(h->tail_out); (void)hipFree(h->counters);
    (void)hipFree(h->tables); (void)hipFree(h->steer); (void)hipFree(h->dev_cnt);
    (void)hipFree(h->ref_pow);
    (void)hipFree(h->x_stage); (void)hipFree(h->y_stage); (void)hipFree(h->opst);
    for (int i = 0; i < 10; ++i) (void)hipFree(h->dev_buf[i]);
    (void)hipFree(h->tdf_w); (void)hipFree(h->tdf_buf); (void)hipFree(h->tdf_P);
    (void)hipFree(h->td_mem); (void)hipFree(h->td_cache[0]); (void)hipFree(h->td_cache[1]); (void)hipFree(h->wpe64);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    for (int i = 0; i < 8; ++i) if (h->group_exec[i]) (void)hipGraphExecDestroy(h->group_exec[i]);
    for (int i = 0; i < 7; ++i) { if (h->side[i]) (void)hipStreamDestroy(h->side[i]); if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int i = 0; i < 10; ++i) if (h->ev_fr[i]) (void)hipEventDestroy(h->ev_fr[i]);
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
    if (wpe_chain(h) && h->chain_buf[6]) DS_HIP(h, hipMemsetAsync(h->chain_buf[6], 0, h->chain_bytes[6], h->stream));
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC)
        for (int i = G_FPREV; i <= G_FIXPREV; ++i)
            if (h->chain_buf[i]) DS_HIP(h, hipMemsetAsync(h->chain_buf[i], 0, h->chain_bytes[i], h->stream));
    if (h->cfg.algo == DS_ALGO_FDGSC)
        for (int i = 16; i <= 18; ++i)                                  // delay_aligned, delay_fbf, last bm_output block (ds_api_gsc_chains.hip)
            if (h->chain_buf[i]) DS_HIP(h, hipMemsetAsync(h->chain_buf[i], 0, h->chain_bytes[i], h->stream));
    h->hist_cur = 0; h->est_used = 0;
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
        h->graph_valid = false;            // a captured graph holds the old buffer's address and batch stride
    }
    DS_HIP(h, hipMemcpyAsync(h->steer, steer, need, hipMemcpyHostToDevice, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    h->steer_set = true;
    h->est_used = 0;                      // adaptivebeamformer.py:70-79: a new look direction restarts frameCount
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
        case DS_PARAM_WPE_FP64: {
            ds_handle* t = wpe_chain(h) ? h->sub[1] : h;
            if (!t || t->cfg.algo != DS_ALGO_WPE) return fail(h, DS_EINVAL, "wpe fp64: DS_ALGO_WPE / DS_ALGO_WPE_TD / DS_ALGO_WPE_MVDR handles only");
            if (t->wpe_started) return fail(h, DS_ESTATE, "wpe fp64 must be set before the first frame (or after ds_reset)");
            int rc = set_device(h); if (rc) return rc;
            DS_HIP(h, hipStreamSynchronize(t->stream));
            // a captured sequence holds the address of the double state (and the kernel that reads it): drop every cached graph of the
            // chain before the block goes away, and make the next sequence warm up with plain launches again
            drop_graphs(h); if (t != h) drop_graphs(t);
            if (value == 0) { (void)hipFree(t->wpe64); t->wpe64 = nullptr; return DS_OK; }
            if (!t->wpe64) {
                const size_t bytes = (size_t)t->cfg.batch * wpe64_ust(t) * sizeof(double);
                DS_HIP(h, hipMalloc((void**)&t->wpe64, bytes));
            }
            DS_HIP(h, ds::launch_wpe64_init(t->wpe64, t->cfg.batch, t->K, (long long)wpe64_ust(t), t->cfg.n_mics, t->filter_len, t->stream));
            DS_HIP(h, hipStreamSynchronize(t->stream));
            return DS_OK;
        }
        case DS_PARAM_WPE_DELAY:
            if (!wpe_chain(h) || value < 0 || value > 64) return fail(h, DS_EINVAL, "wpe delay: chain handles only, 0..64 frames");
            if (h->chain_buf[6]) return fail(h, DS_ESTATE, "wpe delay must be set before the first call");
            h->wpe_delay = value;
            return DS_OK;
        case DS_PARAM_METHOD:
            if (value < 0 || value > 3) return fail(h, DS_EINVAL, "method must be 0..3");
            if (value == DS_METHOD_TFGSC && h->cfg.algo == DS_ALGO_ADAPTIVE && !h->cfg.track_ryy)
                return fail(h, DS_ESTATE, "method TFGSC needs ds_config.track_ryy = 1");
            if (value == DS_METHOD_TFGSC && h->cfg.algo == DS_ALGO_ADAPTIVE_PF)
                return fail(h, DS_EUNSUPPORTED, "method TFGSC: not on a DS_ALGO_ADAPTIVE_PF handle (it keeps no Ryy)");
            if (value != h->method) h->est_used = 0;       // adaptivebeamformer.py:74-79: a new method restarts frameCount
            h->method = value;
            return DS_OK;
        case DS_PARAM_EST_POS:
            if (h->cfg.algo != DS_ALGO_ADAPTIVE && h->cfg.algo != DS_ALGO_ADAPTIVE_PF) return fail(h, DS_EINVAL, "est pos: DS_ALGO_ADAPTIVE / DS_ALGO_ADAPTIVE_PF handles only");
            if (value < -1) return fail(h, DS_EINVAL, "est pos must be -1 (None) or >= 0");
            { const int jr = join_groups(h); if (jr) return jr; }
            h->est_pos = value; h->graph_valid = false;
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
        case DS_PARAM_FDAF_TWO_PATH: h->fdaf_two_path = value != 0; return DS_OK;
        case DS_PARAM_TAIL_ASYNC: {
            if (h->cfg.algo != DS_ALGO_SUBBAND_GSC || !h->side[2]) return fail(h, DS_EINVAL, "tail async: pipelined DS_ALGO_SUBBAND_GSC chain handles only");
            for (int i = 0; i < 24; ++i) if (h->chain_bytes[i]) return fail(h, DS_ESTATE, "tail async must be set before the first call");
            h->tail_async = value != 0;
            return DS_OK;
        }
        case DS_PARAM_REF_POWERS:
            if (h->cfg.algo != DS_ALGO_GSC) return fail(h, DS_EINVAL, "ref powers: DS_ALGO_GSC handles only");
            { const int jr = join_groups(h); if (jr) return jr; }
            h->ref_powers = value != 0;
            h->ref_pow_T = 0;
            return DS_OK;
        case DS_PARAM_POSTFILTER:
            if (h->cfg.algo != DS_ALGO_TDGSC && h->cfg.algo != DS_ALGO_FDGSC) return fail(h, DS_EINVAL, "postfilter: TDGSC / FDGSC chain handles only");
            h->postfilter = value != 0;
            return DS_OK;
        case DS_PARAM_MCSPP_REPEAT:
            if (h->cfg.algo == DS_ALGO_MCSPP_MVDR) {
                const int rc = ds_set_param_i(h->sub[1], id, value);
                if (rc) return fail(h, rc, h->sub[1]->err);
                h->mcspp_repeat = value != 0; h->graph_valid = false;
                return DS_OK;
            }
            if (h->cfg.algo != DS_ALGO_MCSPP) return fail(h, DS_EINVAL, "mcspp repeat: DS_ALGO_MCSPP handles only");
            h->mcspp_repeat = value != 0;
            return DS_OK;
        case DS_PARAM_SPLIT:
            if (value < 1 || value > 8) return fail(h, DS_EINVAL, "split must be 1..8");
            if (wpe_chain(h)) {
                // the groups' copies of the device counters start again from the host mirrors (copies of groups that did not run are stale)
                int rc = set_device(h); if (rc) return rc;
                DS_HIP(h, hipStreamSynchronize(h->stream));
                h->parts = value; h->graph_valid = false;
                rc = sync_dev_cnt(h); if (rc) return rc;
                for (int i = 0; i < 10; ++i) if (h->sub[i]) { rc = sync_dev_cnt(h->sub[i]); if (rc) return fail(h, rc, h->sub[i]->err); }
                return DS_OK;
            }
            h->split = value; h->graph_valid = false;
            return DS_OK;
        default: return fail(h, DS_EINVAL, "unknown int parameter id");
    }
}

int ds_set_window(ds_handle* h, const float* window, int n) {
    if (!h || !window) return fail(h, DS_EINVAL, "ds_set_window: NULL argument");
    if (h->cfg.algo != DS_ALGO_TRANSFORM) return fail(h, DS_ESTATE, "ds_set_window: handle is not a DS_ALGO_TRANSFORM object");
    if (n != h->cfg.nfft) return fail(h, DS_ESHAPE, "ds_set_window: the window must have nfft samples");
    int rc = set_device(h); if (rc) return rc;
    const int N = h->cfg.nfft, NC = N / 2, NSTW = NC == 512 ? 512 : 128;
    double W0 = 0.0;
    for (int i = 0; i < N; ++i) W0 += (double)window[i] * (double)window[i];
    if (!(W0 > 0.0)) return fail(h, DS_EINVAL, "ds_set_window: the window has no energy");
    DS_HIP(h, hipStreamSynchronize(h->stream));
    float* win_dev = reinterpret_cast<float*>(h->tables) + (size_t)(NC + 2) * 2 + (size_t)NSTW * 4;      // Tables<NFFT>::win (ds_tables.hpp)
    DS_HIP(h, hipMemcpy(win_dev, window, (size_t)N * sizeof(float), hipMemcpyHostToDevice));
    h->out_scale = (float)((double)h->cfg.hop / W0);                                                   // transform.py:428,479
    return DS_OK;
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
    if (h->cfg.algo == DS_ALGO_TDGSC || h->cfg.algo == DS_ALGO_FDGSC) {
        if (first != 0 || count != h->cfg.batch) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle processes its whole batch");
        if (stream && (hipStream_t)stream != h->stream) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle runs on its own stream (pass NULL)");
        if (layout != DS_LAYOUT_CHANNELS_SAMPLES) return fail(h, DS_EUNSUPPORTED, "ds_process_device: the TDGSC / FDGSC chains take [B][M][n] input");
        if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
        if (n_samples == 0) return DS_OK;
        const long long cs = x_chan_stride > 0 ? x_chan_stride : n_samples;
        if (h->cfg.algo == DS_ALGO_TDGSC) return tdgsc_run(h, x_dev, x_batch_stride, cs, n_samples, h->postfilter, y_dev, y_batch_stride, nullptr, nullptr, nullptr);
        return fdgsc_run(h, x_dev, x_batch_stride, cs, n_samples, h->postfilter, 1, y_dev, y_batch_stride, nullptr, nullptr, nullptr, nullptr, nullptr,
                         nullptr, nullptr, nullptr);
    }
    if (h->cfg.algo == DS_ALGO_MCSPP_MVDR) {
        if (first != 0 || count != h->cfg.batch) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle processes its whole batch");
        if (stream && (hipStream_t)stream != h->stream) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle runs on its own stream (pass NULL)");
        if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
        if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES) return fail(h, DS_EINVAL, "ds_process_device: unknown layout");
        if (((uintptr_t)x_dev & 15) || ((uintptr_t)y_dev & 15) || (x_batch_stride & 3) || (x_chan_stride & 3) || (y_batch_stride & 3))
            return fail(h, DS_EINVAL, "ds_process_device: device buffers must be 16-byte aligned (strides multiples of 4 elements)");
        if (n_samples == 0) return DS_OK;
        return nbmvdr_process_device(h, x_dev, layout, x_batch_stride, x_chan_stride, n_samples, y_dev, y_batch_stride, nullptr);
    }
    if (wpe_chain(h)) {
        if (first != 0 || count != h->cfg.batch) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle processes its whole batch");
        if (stream && (hipStream_t)stream != h->stream) return fail(h, DS_EUNSUPPORTED, "ds_process_device: a chain handle runs on its own stream (pass NULL)");
        if (!h->steer_set && !h->wpe_only) return fail(h, DS_ESTATE, "ds_process_device: call ds_set_steering first");
        if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
        if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES) return fail(h, DS_EINVAL, "ds_process_device: unknown layout");
        if (n_samples == 0) return DS_OK;
        return chain_process_device(h, x_dev, layout, x_batch_stride, x_chan_stride, n_samples, y_dev, y_batch_stride);
    }
    if (!dsi::frames_algo(h->cfg.algo)) return fail(h, DS_ESTATE, "ds_process_device: this handle is a frame-level object; use ds_stft / ds_*_estimate / ds_sub*_update");
    if (!h->steer_set) return fail(h, DS_ESTATE, "ds_process_device: call ds_set_steering first");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0)
        return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
    if (first < 0 || count < 0 || first + count > h->cfg.batch)
        return fail(h, DS_EINVAL, "ds_process_device: utterance range outside the handle's batch");
    if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES)
        return fail(h, DS_EINVAL, "ds_process_device: unknown layout");
    if (((uintptr_t)x_dev & 15) || ((uintptr_t)y_dev & 15) || (x_batch_stride & 3) || (x_chan_stride & 3))
        return fail(h, DS_EINVAL, "ds_process_device: device buffers must be 16-byte aligned");
    if (n_samples == 0 || count == 0) { if (h->ref_powers) h->ref_pow_T = 0; return DS_OK; }      // (no hop ran: no powers of a 'last call' to read)
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
    if (h->ref_powers) {                                    // DS_PARAM_REF_POWERS: [B][T][K][M] of this call, indexed by the handle's utterance numbers
        if (h->group_enqueue) return fail(h, DS_ESTATE, "ds_process_device: DS_PARAM_REF_POWERS is on — no utterance groups");
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(h, DS_ESTATE, "ds_process_device: DS_PARAM_REF_POWERS is on — no hipGraph capture");
        const size_t need = (size_t)h->cfg.batch * p.T * h->K * h->cfg.n_mics;
        if (need > h->ref_pow_cap) {
            DS_HIP(h, hipStreamSynchronize(s));
            (void)hipFree(h->ref_pow); h->ref_pow = nullptr; h->ref_pow_cap = 0;
            DS_HIP(h, hipMalloc((void**)&h->ref_pow, need * sizeof(float)));
            h->ref_pow_cap = need;
        }
        if (count != h->cfg.batch) DS_HIP(h, hipMemsetAsync(h->ref_pow, 0, need * sizeof(float), s));   // utterances outside the range read as zeros
        p.ref_pow = h->ref_pow;
        h->ref_pow_T = p.T;
        h->ref_pow_stream = s;                              // ds_get_state(DS_FIELD_REF_POWERS) waits for THIS stream (a caller's, possibly)
    }
    // calls of several hops: the hop-pipelined kernel (same results bit for bit; at one or two hops per call it has nothing to overlap)
    // ... or the build of the kernel for long calls, where there is one (same results bit for bit)
    const ds::launch_fn launch = (h->ki.launch_pipe && p.T >= h->pipe_min_T && !h->ref_powers) ? h->ki.launch_pipe
                                 : (h->ki.launch_long && p.T >= DS_LONG_MIN_T) ? h->ki.launch_long : h->ki.launch;
    if (h->est_pos >= 0 && (h->cfg.algo == DS_ALGO_ADAPTIVE || h->cfg.algo == DS_ALGO_ADAPTIVE_PF)) {
        // adaptivebeamfomer.estPos (adaptivebeamformer.py:90-93): the call as up to three launches — the frames whose every bin still updates Rvv
        // (gate always open), the ONE frame in which the slot count runs out (its leading bins only), the frames behind it (gate shut)
        if (first != 0 || count != h->cfg.batch || h->group_enqueue) return fail(h, DS_EUNSUPPORTED, "ds_process_device: DS_PARAM_EST_POS is on — the whole batch per call, no utterance groups");
        if (h->ref_powers) return fail(h, DS_EUNSUPPORTED, "ds_process_device: DS_PARAM_EST_POS and DS_PARAM_REF_POWERS exclude each other");
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(h, DS_ESTATE, "ds_process_device: DS_PARAM_EST_POS is on — no hipGraph capture");
        const long long K = h->K, left = h->est_pos > h->est_used ? h->est_pos - h->est_used : 0;      // (estPos lowered below the count so far: nothing left)
        const int T = p.T, hop = h->cfg.hop;
        const int n_full = (int)(left / K < T ? left / K : T);
        const int n_part = (n_full < T && left - (long long)n_full * K > 0) ? 1 : 0;
        const long long xs = layout == DS_LAYOUT_CHANNELS_SAMPLES ? hop : (long long)hop * h->cfg.n_mics;
        int t0 = 0;
        auto seg = [&](int frames, float gate, unsigned kinv) -> hipError_t {
            if (frames <= 0) return hipSuccess;
            Params q = p;
            q.x = x_dev + (long long)t0 * xs; q.y = y_dev + (long long)t0 * hop; q.T = frames; q.gate = gate; q.gate_kinv = kinv;
            t0 += frames;
            return ((h->ki.launch_long && frames >= DS_LONG_MIN_T) ? h->ki.launch_long : h->ki.launch)(q, count, s);
        };
        DS_HIP(h, seg(n_full, 2.0f, 0u));                                        // mcra.p <= 0.999 < 2: every bin updates
        DS_HIP(h, seg(n_part, 2.0f, ~(unsigned)(left - (long long)n_full * K))); // bins [0, r)
        DS_HIP(h, seg(T - n_full - n_part, -1.0f, 0u));                          // mcra.p >= 0.001 > -1: no bin updates
        h->est_used += (long long)n_full * K + (n_part ? left - (long long)n_full * K : 0);
        return DS_OK;
    }
    DS_HIP(h, launch(p, count, s));
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
    if (h->ref_powers && (n_calls > 1 || graph != 0))
        return fail(h, DS_ESTATE, "ds_process_device_seq: DS_PARAM_REF_POWERS keeps the powers of ONE plain call (n_calls 1, graph 0)");
    if (h->est_pos >= 0) graph = 0;                           // DS_PARAM_EST_POS: the slot count lives on the host, every call is laid out for it
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    const bool chain = wpe_chain(h) || h->cfg.algo == DS_ALGO_SUBBAND_GSC || h->cfg.algo == DS_ALGO_MCSPP_MVDR;
    if (h->cfg.algo == DS_ALGO_TDGSC || h->cfg.algo == DS_ALGO_FDGSC) graph = 0;      // their stages keep the frame counters on the host
    // the SubbandGSC chain pipelines its stages over the enqueued blocks on up to five streams; replayed as ONE hipGraph the branches are
    // serialised by the graph executor (measured 0.34-0.50 ms per block against 0.26 ms with plain launches), so the sequence is launched
    // plainly.  DS_CHAIN_SERIAL_FRONT=1 (every stage behind the previous one) keeps the replay path
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC && h->front_async && graph == 1) graph = 0;
    if (chain && graph != 0) {
        // a chain replays as a graph once this call shape has run with plain launches (every stage buffer sized, nothing left to
        // allocate or synchronise inside the capture) and the stages' host-side start-up branches are behind (McSpp's first frames)
        const bool ready = h->chain_warm_n == n_samples_per_call && (h->cfg.algo != DS_ALGO_SUBBAND_GSC || h->sub[2]->op_frm >= 5) &&
                           (!stream || (hipStream_t)stream == h->stream);
        if (!ready) { if (graph == 2) return DS_OK; graph = 0; }
    }
    // Fused frame kernels, DS_PARAM_SPLIT = S > 1: the utterance range as S groups, each on its own stream at its own pace (group 0: the
    // handle's stream, group g: side[g - 1]) — every group replays its own hipGraph of the sequence (or launches it plainly), nothing joins
    // the groups between calls (join_groups() does when anything else touches the handle).  While one group's kernel is in its launch gap
    // or its last round of workgroups the other group's kernel fills the CUs: +12..15 % at 2048-4096 utterances per call, +3 % at 16 384,
    // nothing at 1024 (one round of workgroups).  A caller-provided stream keeps everything on that stream.
    const bool frames = dsi::frames_algo(h->cfg.algo);
    const int ng = (frames && !stream && !h->ref_powers && h->est_pos < 0) ? (h->split < count ? h->split : (count > 0 ? count : 1)) : 1;
    if (ng > 1) {
        DS_HIP(h, hipSetDevice(h->device));                 // not set_device(): the groups stay on their streams between calls
        if (!h->ev_fork) DS_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        for (int g = 0; g < ng - 1; ++g) {
            if (!h->side[g]) DS_HIP(h, hipStreamCreateWithFlags(&h->side[g], hipStreamNonBlocking));
            if (!h->ev_join[g]) DS_HIP(h, hipEventCreateWithFlags(&h->ev_join[g], hipEventDisableTiming));
        }
        auto range = [&](int g, int& lo, int& hi) { lo = (int)((long long)count * g / ng); hi = (int)((long long)count * (g + 1) / ng); };
        auto enqueue = [&](int g, hipStream_t sg) {         // the n_calls launches of group g
            int lo, hi;
            range(g, lo, hi);
            h->group_enqueue = true;                        // ds_process_device -> set_device() would join the very groups being launched
            int rc = DS_OK;
            for (int i = 0; i < n_calls && rc == DS_OK; ++i)
                rc = ds_process_device(h, x_dev + (long long)i * x_call_stride + (long long)lo * x_batch_stride, layout, x_batch_stride,
                                       x_chan_stride, n_samples_per_call, y_dev + (long long)i * y_call_stride + (long long)lo * y_batch_stride,
                                       y_batch_stride, first + lo, hi - lo, (void*)sg);
            h->group_enqueue = false;
            return rc;
        };
        if (graph != 0) {
            float fl[6] = {h->alpha_y, h->alpha_v, h->diag, h->gate, h->mu, 0.0f};
            long long fbits[3];
            std::memcpy(fbits, fl, sizeof fbits);
            const long long key[16] = {(long long)(uintptr_t)x_dev, (long long)(uintptr_t)y_dev, layout, x_batch_stride, x_chan_stride,
                                       x_call_stride, n_samples_per_call, n_calls, y_batch_stride, y_call_stride,
                                       ((long long)first << 32) | (unsigned)count, ((long long)h->method << 32) | (unsigned)h->mcra_L,
                                       fbits[0], fbits[1], fbits[2] ^ ((long long)ng << 40) ^ (1LL << 50), (long long)(uintptr_t)h->steer};
            if (!h->graph_valid || std::memcmp(key, h->graph_key, sizeof key) != 0) {
                { const int jr = join_groups(h); if (jr) return jr; }
                DS_HIP(h, hipStreamSynchronize(h->stream));
                h->graph_valid = false;
                if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
                for (int g = 0; g < 8; ++g) if (h->group_exec[g]) { (void)hipGraphExecDestroy(h->group_exec[g]); h->group_exec[g] = nullptr; }
                for (int g = 0; g < ng; ++g) {
                    hipStream_t sg = g == 0 ? h->stream : h->side[g - 1];
                    DS_HIP(h, hipStreamBeginCapture(sg, hipStreamCaptureModeThreadLocal));
                    const int crc = enqueue(g, sg);
                    hipGraph_t gr = nullptr;
                    const hipError_t e = hipStreamEndCapture(sg, &gr);
                    if (crc != DS_OK) { if (gr) (void)hipGraphDestroy(gr); return crc; }
                    if (e != hipSuccess) return fail(h, DS_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                    const hipError_t e2 = hipGraphInstantiate(&h->group_exec[g], gr, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(gr);
                    if (e2 != hipSuccess) return fail(h, DS_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2));
                }
                std::memcpy(h->graph_key, key, sizeof key);
                h->graph_valid = true;
            }
            if (graph == 2) return DS_OK;
        }
        if (!h->groups_open) {                              // first sequence since the groups were joined: the side streams start from the handle's
            DS_HIP(h, hipEventRecord(h->ev_fork, h->stream));
            for (int g = 1; g < ng; ++g) DS_HIP(h, hipStreamWaitEvent(h->side[g - 1], h->ev_fork, 0));
        }
        for (int g = 0; g < ng; ++g) {
            hipStream_t sg = g == 0 ? h->stream : h->side[g - 1];
            if (graph != 0) DS_HIP(h, hipGraphLaunch(h->group_exec[g], sg));
            else { const int rc = enqueue(g, sg); if (rc) return rc; }
        }
        h->groups_open = true;
        return DS_OK;
    }
    if (graph == 0) {
        if (chain) h->chain_warm_n = n_samples_per_call;
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
                               fbits[0], fbits[1], fbits[2] ^ ((long long)h->split << 40) ^ ((long long)h->parts << 44) ^ ((long long)((wpe_chain(h) && h->sub[1] && h->sub[1]->wpe64) ? 1 : 0) << 52),
                               (long long)(uintptr_t)h->steer};
    if (!h->graph_valid || std::memcmp(key, h->graph_key, sizeof key) != 0) {
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
        h->graph_valid = false;
        hipStream_t cs = h->stream;                       // capture on the handle's own stream
        DS_HIP(h, hipStreamSynchronize(cs));
        ChainMirrors before;
        if (chain) { mirrors_get(h, before); const int jr = join_groups(h); if (jr) return jr; }
        h->front_open = false;                              // a chain's front-end stream joins the capture from the chain's stream
        DS_HIP(h, hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int crc = DS_OK;
        for (int i = 0; i < n_calls && crc == DS_OK; ++i)
            crc = ds_process_device(h, x_dev + (long long)i * x_call_stride, layout, x_batch_stride, x_chan_stride, n_samples_per_call,
                                    y_dev + (long long)i * y_call_stride, y_batch_stride, first, count, (void*)cs);
        if (chain && crc == DS_OK) crc = join_groups(h);    // utterance groups of a chain: the side streams come back before the capture ends
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(cs, &g);
        h->front_open = false;
        if (chain) {                                      // nothing ran: the mirrors go back, one replay advances them by what the capture did
            ChainMirrors after;
            mirrors_get(h, after);
            for (int i = 0; i < 11; ++i) { h->adv_frames[i] = after.frm[i] - before.frm[i]; h->adv_td[i] = after.td[i] ^ before.td[i]; }
            const int dl = h->wpe_delay > 0 ? h->wpe_delay : 1;
            h->adv_hist = ((after.hist - before.hist) % dl + dl) % dl;
            mirrors_set(h, before);
        }
        if (crc != DS_OK) { if (g) (void)hipGraphDestroy(g); return crc; }
        if (e != hipSuccess) return fail(h, DS_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(h, DS_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        std::memcpy(h->graph_key, key, sizeof key);
        h->graph_valid = true;
    }
    if (graph == 2) return DS_OK;
    if (chain) { const int jr = join_groups(h); if (jr) return jr; }
    DS_HIP(h, hipGraphLaunch(h->graph_exec, s));
    h->front_open = false;                                  // whatever follows on the front-end stream follows the replay
    if (chain) {
        for (int i = 0; i < 11; ++i) {
            ds_handle* t = i < 10 ? h->sub[i] : h;
            if (!t) continue;
            advance_host_counters_keep_first(t, h->adv_frames[i], t->cfg.algo == DS_ALGO_MCSPP ? 65 : t->mcra_L);
            if (t->op == ds::OP_WPE && h->adv_frames[i] > 0) t->wpe_started = true;   // a replay never passes through wpe_launch, which sets it
            t->td_cur ^= h->adv_td[i];
        }
        h->hist_cur = (h->hist_cur + h->adv_hist) % (h->wpe_delay > 0 ? h->wpe_delay : 1);
    }
    return DS_OK;
}

static int process_host(ds_handle* h, const float* x, int layout, int n_samples, void* y, bool f64);
int ds_process(ds_handle* h, const float* x, int layout, int n_samples, float* y) { return process_host(h, x, layout, n_samples, y, false); }
// ... with the enhanced samples widened to float64 on the device (what the reference's process() returns; exact)
int ds_process_f64(ds_handle* h, const float* x, int layout, int n_samples, double* y) { return process_host(h, x, layout, n_samples, y, true); }
static int process_host(ds_handle* h, const float* x, int layout, int n_samples, void* y, bool f64) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_process: NULL argument");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0)
        return fail(h, DS_ESHAPE, "ds_process: n_samples must be a multiple of hop");
    if (n_samples == 0) { if (h->ref_powers) h->ref_pow_T = 0; return DS_OK; }
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
    // (round 6: the batch as utterance groups — upload | kernel | download as a pipeline over three streams — was built and measured: 4.8 M
    // frames/s against 9.4 M for this plain sequence at B = 1024 and 4 hops per call, every group count from 2 to 7 slower than one: a
    // hipMemcpyAsync from pageable memory pays its fixed part per call, and ONE 16 MB upload already runs at 0.88 of the pinned rate;
    // profiles/r06a/host_api_io_groups_ab.txt.  Removed.  What a call does lose is the first touch of a freshly allocated output array: the
    // Python layer hands its outputs out of a pool of page-locked blocks, ds_host_alloc)
    DS_HIP(h, hipMemcpyAsync(h->x_stage, x, xe * sizeof(float), hipMemcpyHostToDevice, h->stream));
    h->front_open = false;                                  // a chain's front-end stream follows the upload
    rc = ds_process_device(h, h->x_stage, layout, (long long)(M * (size_t)n_samples), 0, n_samples, h->y_stage,
                           (long long)n_samples, 0, h->cfg.batch, nullptr);
    if (rc) return rc;
    rc = join_groups(h); if (rc) return rc;
    if (f64) {
        rc = stage_reserve(h, 6, ye * sizeof(double)); if (rc) return rc;
        DS_HIP(h, ds::launch_float_to_double(h->y_stage, (double*)h->dev_buf[6], (long long)ye, h->stream));
        DS_HIP(h, hipMemcpyAsync(y, h->dev_buf[6], ye * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else {
        DS_HIP(h, hipMemcpyAsync(y, h->y_stage, ye * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

// ---- frame-level entry points ---------------------------------------------------------------------
int ds_process_pcm16(ds_handle* h, const int16_t* pcm, int n_total_channels, int first_channel, int n_samples, int16_t* out) {
    if (!h || !pcm || !out) return fail(h, DS_EINVAL, "ds_process_pcm16: NULL argument");
    if (!dsi::frames_algo(h->cfg.algo)) return fail(h, DS_ESTATE, "ds_process_pcm16: handle is a frame-level object");
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

// page-locked host memory for the caller's buffers (hipHostMalloc): a download into it needs no page to be faulted in or pinned first — the Python
// layer hands its output arrays out of a pool of such blocks (distantspeech_amd/engine.py)
void* ds_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
int ds_host_free(void* p) {
    if (!p) return DS_OK;
    return hipHostFree(p) == hipSuccess ? DS_OK : DS_EHIP;
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
    const bool pf = h->cfg.algo == DS_ALGO_ADAPTIVE_PF;
    const bool ad = h->cfg.algo == DS_ALGO_ADAPTIVE || pf, gsc = h->cfg.algo == DS_ALGO_GSC;
    switch (field) {
        case DS_FIELD_RVV: return ad ? B * K * M * M * 2 * sizeof(float) : 0;
        case DS_FIELD_RYY: return (ad && !pf && h->cfg.track_ryy) ? B * K * M * M * 2 * sizeof(float) : 0;
        case DS_FIELD_MCRA_S: case DS_FIELD_MCRA_SMIN: case DS_FIELD_MCRA_STMP: case DS_FIELD_MCRA_P:
        case DS_FIELD_MCRA_LAMBDA_D: return ad ? B * K * sizeof(float) : 0;
        case DS_FIELD_PHI_YY: case DS_FIELD_PHI_VV: return (gsc || pf) ? B * K * M * M * sizeof(float) : 0;
        case DS_FIELD_G_AIC: return gsc ? B * K * (M - 1) * 2 * sizeof(float) : 0;
        case DS_FIELD_STFT_TAIL: return tail_in_bytes(h);
        case DS_FIELD_OLA_TAIL: return tail_out_bytes(h);
        case DS_FIELD_COUNTERS: return counters_bytes(h);
        case DS_FIELD_OP_STATE:
            if (h->tdf_w) return (size_t)h->cfg.batch * h->cfg.filter_len * sizeof(float);
            return opst_bytes(h);
        case DS_FIELD_NOTCH_MEM: return h->td_mem ? B * M * 2 * sizeof(float) : 0;
        case DS_FIELD_WPE_STATE64: return wpe64_bytes(h);
        case DS_FIELD_H: return (ad && h->method != DS_METHOD_TFGSC) ? B * K * M * 2 * sizeof(float) : 0;
        case DS_FIELD_REF_POWERS: return (gsc && h->ref_powers) ? B * (size_t)h->ref_pow_T * K * M * sizeof(float) : 0;
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
    if (field == DS_FIELD_NOTCH_MEM) {                      // carried in double on the device; the field is the float32 view of it
        std::vector<double> m(need / sizeof(float));
        DS_HIP(h, hipMemcpy(m.data(), h->td_mem, m.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < m.size(); ++i) ((float*)dst)[i] = (float)m[i];
        return DS_OK;
    }
    if (field == DS_FIELD_WPE_STATE64) { DS_HIP(h, hipMemcpy(dst, h->wpe64, need, hipMemcpyDeviceToHost)); return DS_OK; }
    if (field == DS_FIELD_REF_POWERS) {
        if (h->ref_pow_stream && h->ref_pow_stream != h->stream) DS_HIP(h, hipStreamSynchronize(h->ref_pow_stream));   // the launch that wrote them
        DS_HIP(h, hipMemcpy(dst, h->ref_pow, need, hipMemcpyDeviceToHost));
        return DS_OK;
    }
    if (field == DS_FIELD_OP_STATE) {
        DS_HIP(h, hipMemcpy(dst, h->tdf_w ? (const void*)h->tdf_w : (const void*)h->opst, need, hipMemcpyDeviceToHost));
        return DS_OK;
    }
    if (field == DS_FIELD_H) {
        if (!h->steer_set) return fail(h, DS_ESTATE, "ds_get_state(DS_FIELD_H): call ds_set_steering first");
        float* Hd = nullptr;
        DS_HIP(h, hipMalloc((void**)&Hd, need));
        hipError_t e = ds::launch_mvdr_probe(h->cfg.n_mics, reinterpret_cast<const float*>(h->bins), (long long)bins_ust(h), h->KP, h->ki.NF, h->cfg.batch, h->K,
                                             reinterpret_cast<const float*>(h->steer), h->steer_per_utt ? (long long)h->K * h->cfg.n_mics : 0, h->diag, h->method, Hd, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dst, Hd, need, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        (void)hipFree(Hd);
        if (e != hipSuccess) return fail(h, DS_EHIP, std::string("ds_get_state(DS_FIELD_H): ") + hipGetErrorString(e));
        return DS_OK;
    }
    // per-bin fields: pull the raw planes and unpack on the host
    std::vector<float> raw(bins_bytes(h) / sizeof(float));
    DS_HIP(h, hipMemcpy(raw.data(), h->bins, bins_bytes(h), hipMemcpyDeviceToHost));
    const int B = h->cfg.batch, K = h->K, M = h->cfg.n_mics, KP = h->KP, NF = h->ki.NF, NPF = NF / 4, RT = NF % 4;
    auto at = [&](int b, int k, int f) -> float {   // float f of bin k of utterance b: NPF float4 planes, then the narrow plane [KP][RT]
        const size_t u = (size_t)b * bins_ust(h);
        return f < 4 * NPF ? raw[u + ((size_t)(f / 4) * KP + k) * 4 + (f % 4)] : raw[u + (size_t)NPF * KP * 4 + (size_t)k * RT + (f - 4 * NPF)];
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
        case DS_FIELD_PHI_YY: sym(h->cfg.algo == DS_ALGO_ADAPTIVE_PF ? M * M + 5 : 0); break;                       // StateLayout::PYY / PF_PYY
        case DS_FIELD_PHI_VV: sym((h->cfg.algo == DS_ALGO_ADAPTIVE_PF ? M * M + 5 : 0) + M * (M + 1) / 2); break;
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

// state kept outside the common buffers: the front end's notch memories and FIR history, the sample-wise filters' weights / buffer / P
struct ExtraState { void* ptr; size_t bytes; };
static int extra_state(const ds_handle* h, ExtraState out[3]) {
    int n = 0;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, Lf = h->cfg.filter_len;
    if (h->cfg.algo == DS_ALGO_FRONTEND) {
        out[n++] = {h->td_mem, B * M * 2 * sizeof(double)};
        if (h->td_L > 1) out[n++] = {h->td_cache[h->td_cur], B * (size_t)(h->td_L - 1) * M * sizeof(float)};
    }
    if (h->wpe64) out[n++] = {h->wpe64, wpe64_bytes(h)};
    if (h->tdf_w) {
        out[n++] = {h->tdf_w, B * Lf * sizeof(float)};
        out[n++] = {h->tdf_buf, B * Lf * sizeof(float)};
        if (h->tdf_P) out[n++] = {h->tdf_P, B * Lf * Lf * sizeof(float)};
    }
    return n;
}
// every handle's section of a checkpoint blob starts with this header; ds_import_state refuses a blob written for another configuration
// layout: bumped whenever the serialised arrangement of the state changes (2 = per-bin planes without spare words, 128-byte utterance
// stride, foreground filter in the FDAF state); modes: the settings that shape the state or its meaning (two-path FDAF, McSpp repeat,
// shared-reference / complemented-p subband filters); out_scale_bits: hop / sum(window^2), which moves with a caller-supplied window
struct BlobHeader { uint32_t magic, version; int32_t algo, nfft, hop, n_mics, batch, filter_len, td_L, track_ryy, layout, modes, wpe_delay; uint32_t out_scale_bits; };
static const uint32_t BLOB_MAGIC = 0x44534348u;      // "DSCH"
static const int32_t BLOB_LAYOUT = DS_STATE_LAYOUT;      // 3: operator state as float4 planes [b][f / 4][k][f % 4], FIR history channel-major [b][m][L - 1]
                                                         // 4: plane rows of K rounded up to 8 lanes (ds_core.hpp plane_len); RLS-WPE blocks on 128-byte lines (ds_wpe.hpp wpe_layout)
static BlobHeader blob_header(const ds_handle* h) {
    uint32_t osb;
    std::memcpy(&osb, &h->out_scale, sizeof osb);
    const int32_t modes = (h->fdaf_two_path ? 1 : 0) | (h->mcspp_repeat ? 2 : 0) | (h->x_fan > 1 ? 4 : 0) | (h->p_complement ? 8 : 0) | (h->wpe64 ? 16 : 0);
    return BlobHeader{BLOB_MAGIC, (uint32_t)DS_VERSION, h->cfg.algo, h->cfg.nfft, h->cfg.hop, h->cfg.n_mics, h->cfg.batch, h->filter_len, h->td_L,
                      h->cfg.track_ryy, BLOB_LAYOUT, modes, h->wpe_delay, osb};
}
static size_t own_state_bytes(const ds_handle* h) {
    ExtraState ex[3];
    size_t n = sizeof(BlobHeader) + bins_bytes(h) + tail_in_bytes(h) + tail_out_bytes(h) + counters_bytes(h) + opst_bytes(h) + 4 * sizeof(int);
    for (int i = 0, k = extra_state(h, ex); i < k; ++i) n += ex[i].bytes;
    return n;
}
static size_t fdgsc_slot_bytes(const ds_handle* h, int i) {            // chain_buf[16 + i]: delay_aligned [B][M][hop/2], delay_fbf [B][hop], bm_last [B][M][hop]
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, FL = h->cfg.hop;
    return i == 0 ? B * M * (FL / 2) * 4 : i == 1 ? B * FL * 4 : B * M * FL * 4;
}
static size_t chain_hist_bytes(const ds_handle* h) {
    if (h->cfg.algo == DS_ALGO_FDGSC) return fdgsc_slot_bytes(h, 0) + fdgsc_slot_bytes(h, 1) + fdgsc_slot_bytes(h, 2);
    if (h->cfg.algo == DS_ALGO_SUBBAND_GSC) return (size_t)h->cfg.batch * (h->K * 8 + h->cfg.hop * 4);
    return wpe_chain(h) ? (size_t)h->cfg.batch * (h->wpe_delay > 0 ? h->wpe_delay : 1) * h->K * h->cfg.n_mics * 8 : 0;
}

size_t ds_state_bytes(const ds_handle* h) {
    if (!h) return 0;
    size_t n = own_state_bytes(h) + chain_hist_bytes(h);
    for (int i = 0; i < 10; ++i) if (h->sub[i]) n += ds_state_bytes(h->sub[i]);
    return n;
}

// the same without the blob's framing (header + uniform counters of every handle in it) and without the padding of the plane arrays (the
// lanes K .. KP - 1 of every row, the rounding of the floats per bin to whole float4 planes, the line padding of the RLS-WPE blocks): the
// bytes that carry state — what one call must move each way
static size_t plane_padding_bytes(const ds_handle* h) {
    const size_t B = h->cfg.batch, K = h->K;
    size_t pad = 0;
    if (bins_bytes(h)) pad += bins_bytes(h) - B * (size_t)h->ki.NF * K * sizeof(float);
    if (h->op >= 0 && h->NF > 0 && opst_bytes(h)) {
        const size_t live = h->cfg.algo == DS_ALGO_WPE ? B * K * (size_t)ds::wpe_bin_floats_packed(h->cfg.n_mics, h->filter_len) * sizeof(float)
                                                        : B * (size_t)h->NF * K * sizeof(float);
        if (live < opst_bytes(h)) pad += opst_bytes(h) - live;
    }
    return pad;
}
size_t ds_state_payload_bytes(const ds_handle* h) {
    if (!h) return 0;
    size_t n = own_state_bytes(h) - sizeof(BlobHeader) - 4 * sizeof(int) + chain_hist_bytes(h) - plane_padding_bytes(h);
    for (int i = 0; i < 10; ++i) if (h->sub[i]) n += ds_state_payload_bytes(h->sub[i]);
    return n;
}

int ds_chain_stage_info(const ds_handle* h, int i, int32_t* algo, int32_t* n_mics, int32_t* batch, size_t* payload_bytes) {
    if (!h || i < 0 || i >= 10 || !h->sub[i]) return DS_EINVAL;
    const ds_handle* s = h->sub[i];
    if (algo) *algo = s->cfg.algo;
    if (n_mics) *n_mics = s->cfg.n_mics;
    if (batch) *batch = s->cfg.batch;
    if (payload_bytes) *payload_bytes = ds_state_payload_bytes(s);
    return DS_OK;
}

size_t ds_chain_stage_field_bytes(const ds_handle* h, int i, int field) {
    if (!h || i < 0 || i >= 10 || !h->sub[i]) return 0;
    return ds_field_bytes(h->sub[i], field);
}

int ds_chain_stage_state(ds_handle* h, int i, int field, void* dst, size_t bytes) {
    if (!h || i < 0 || i >= 10 || !h->sub[i]) return fail(h, DS_EINVAL, "ds_chain_stage_state: the handle has no such stage");
    int rc = set_device(h); if (rc) return rc;                          // (joins the chain's utterance groups)
    DS_HIP(h, hipStreamSynchronize(h->stream));
    rc = ds_get_state(h->sub[i], field, dst, bytes);
    return rc ? fail(h, rc, h->sub[i]->err) : DS_OK;
}

int ds_export_state(ds_handle* h, void* dst, size_t bytes) {
    if (!h || !dst) return fail(h, DS_EINVAL, "ds_export_state: NULL argument");
    if (bytes != ds_state_bytes(h)) return fail(h, DS_ESHAPE, "ds_export_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    char* d = (char*)dst;
    { const BlobHeader hd = blob_header(h); std::memcpy(d, &hd, sizeof hd); d += sizeof hd; }
    if (bins_bytes(h)) DS_HIP(h, hipMemcpy(d, h->bins, bins_bytes(h), hipMemcpyDeviceToHost));
    d += bins_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->tail_in, tail_in_bytes(h), hipMemcpyDeviceToHost)); d += tail_in_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->tail_out, tail_out_bytes(h), hipMemcpyDeviceToHost)); d += tail_out_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->counters, counters_bytes(h), hipMemcpyDeviceToHost)); d += counters_bytes(h);
    if (opst_bytes(h)) DS_HIP(h, hipMemcpy(d, h->opst, opst_bytes(h), hipMemcpyDeviceToHost));
    d += opst_bytes(h);
    const int uc[4] = {dsi::frames_algo(h->cfg.algo) ? (int)h->est_used : h->op_frm, h->op_ell, h->op_first, h->hist_cur};   // (a frame-kernel handle has no operator counters: the slot carries the estPos count)
    std::memcpy(d, uc, sizeof uc);
    d += sizeof uc;
    {
        ExtraState ex[3];
        for (int i = 0, k = extra_state(h, ex); i < k; ++i) { DS_HIP(h, hipMemcpy(d, ex[i].ptr, ex[i].bytes, hipMemcpyDeviceToHost)); d += ex[i].bytes; }
    }
    for (int i = 0; i < 10; ++i)
        if (h->sub[i]) {
            const size_t n = ds_state_bytes(h->sub[i]);
            rc = ds_export_state(h->sub[i], d, n); if (rc) return fail(h, rc, h->sub[i]->err);
            d += n;
        }
    if (h->cfg.algo == DS_ALGO_FDGSC) {
        for (int i = 0; i < 3; ++i) {                                   // never-run handle: the delays are silence
            const size_t nb = fdgsc_slot_bytes(h, i);
            if (h->chain_buf[16 + i]) DS_HIP(h, hipMemcpy(d, h->chain_buf[16 + i], nb, hipMemcpyDeviceToHost));
            else std::memset(d, 0, nb);
            d += nb;
        }
    } else if (h->cfg.algo == DS_ALGO_SUBBAND_GSC) {
        rc = chain2_reserve(h, h->cfg.hop); if (rc) return rc;
        const size_t n13 = (size_t)h->cfg.batch * h->K * 8, n14 = (size_t)h->cfg.batch * h->cfg.hop * 4;
        DS_HIP(h, hipMemcpy(d, h->chain_buf[G_FPREV], n13, hipMemcpyDeviceToHost));
        DS_HIP(h, hipMemcpy(d + n13, h->chain_buf[G_FIXPREV], n14, hipMemcpyDeviceToHost));
    } else if (chain_hist_bytes(h)) {
        rc = chain_reserve(h, 1); if (rc) return rc;
        DS_HIP(h, hipMemcpy(d, h->chain_buf[6], chain_hist_bytes(h), hipMemcpyDeviceToHost));
    }
    return DS_OK;
}

int ds_import_state(ds_handle* h, const void* src, size_t bytes) {
    if (!h || !src) return fail(h, DS_EINVAL, "ds_import_state: NULL argument");
    if (bytes < sizeof(BlobHeader)) return fail(h, DS_ESHAPE, "ds_import_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    const char* s = (const char*)src;
    {
        BlobHeader got;
        const BlobHeader want = blob_header(h);
        std::memcpy(&got, s, sizeof got);
        if (got.magic != BLOB_MAGIC) return fail(h, DS_EINVAL, "ds_import_state: not a dsenh checkpoint (bad magic)");
        if (std::memcmp(&got, &want, sizeof got) != 0) {
            char buf[768];
            snprintf(buf, sizeof buf, "ds_import_state: checkpoint was written for version %u algo %d nfft %d hop %d mics %d batch %d taps %d fir %d ryy %d "
                     "layout %d modes %d wpe_delay %d window-scale %08x, this handle is version %u algo %d nfft %d hop %d mics %d batch %d taps %d fir %d "
                     "ryy %d layout %d modes %d wpe_delay %d window-scale %08x",
                     got.version, got.algo, got.nfft, got.hop, got.n_mics, got.batch, got.filter_len, got.td_L, got.track_ryy, got.layout, got.modes,
                     got.wpe_delay, got.out_scale_bits,
                     want.version, want.algo, want.nfft, want.hop, want.n_mics, want.batch, want.filter_len, want.td_L, want.track_ryy, want.layout,
                     want.modes, want.wpe_delay, want.out_scale_bits);
            return fail(h, DS_ESHAPE, buf);
        }
        s += sizeof got;
    }
    if (bytes != ds_state_bytes(h)) return fail(h, DS_ESHAPE, "ds_import_state: byte size mismatch");
    if (bins_bytes(h)) DS_HIP(h, hipMemcpy(h->bins, s, bins_bytes(h), hipMemcpyHostToDevice));
    s += bins_bytes(h);
    DS_HIP(h, hipMemcpy(h->tail_in, s, tail_in_bytes(h), hipMemcpyHostToDevice)); s += tail_in_bytes(h);
    DS_HIP(h, hipMemcpy(h->tail_out, s, tail_out_bytes(h), hipMemcpyHostToDevice)); s += tail_out_bytes(h);
    DS_HIP(h, hipMemcpy(h->counters, s, counters_bytes(h), hipMemcpyHostToDevice)); s += counters_bytes(h);
    if (opst_bytes(h)) DS_HIP(h, hipMemcpy(h->opst, s, opst_bytes(h), hipMemcpyHostToDevice));
    // the WPE recursion never touches Im(P_ii) (it is +0 from the initial state on and the downdate keeps it there): a blob that carries
    // anything else on the diagonal would keep it for good, so an imported state is cleaned once here, not every frame in the kernel
    if (h->op == ds::OP_WPE && opst_bytes(h))
        DS_HIP(h, ds::launch_wpe_fix_diag(h->opst, h->cfg.batch, h->K, (long long)op_ust(h), h->cfg.n_mics, h->filter_len, h->stream));
    s += opst_bytes(h);
    int uc[4];
    std::memcpy(uc, s, sizeof uc);
    s += sizeof uc;
    {
        ExtraState ex[3];
        for (int i = 0, k = extra_state(h, ex); i < k; ++i) { DS_HIP(h, hipMemcpy(ex[i].ptr, s, ex[i].bytes, hipMemcpyHostToDevice)); s += ex[i].bytes; }
    }
    if (dsi::frames_algo(h->cfg.algo)) h->est_used = uc[0]; else h->op_frm = uc[0];
    h->op_ell = uc[1]; h->op_first = uc[2]; h->hist_cur = uc[3];
    h->wpe_started = true;                                  // an imported stream is a started one
    rc = sync_dev_cnt(h); if (rc) return rc;
    h->graph_valid = false;
    for (int i = 0; i < 10; ++i)
        if (h->sub[i]) {
            const size_t n = ds_state_bytes(h->sub[i]);
            rc = ds_import_state(h->sub[i], s, n); if (rc) return fail(h, rc, h->sub[i]->err);
            s += n;
        }
    if (h->cfg.algo == DS_ALGO_FDGSC) {
        for (int i = 0; i < 3; ++i) {
            const size_t nb = fdgsc_slot_bytes(h, i);
            if (!h->chain_buf[16 + i]) { DS_HIP(h, hipMalloc((void**)&h->chain_buf[16 + i], nb)); h->chain_bytes[16 + i] = nb; }
            DS_HIP(h, hipMemcpy(h->chain_buf[16 + i], s, nb, hipMemcpyHostToDevice));
            s += nb;
        }
    } else if (h->cfg.algo == DS_ALGO_SUBBAND_GSC) {
        rc = chain2_reserve(h, h->cfg.hop); if (rc) return rc;
        const size_t n13 = (size_t)h->cfg.batch * h->K * 8, n14 = (size_t)h->cfg.batch * h->cfg.hop * 4;
        DS_HIP(h, hipMemcpy(h->chain_buf[G_FPREV], s, n13, hipMemcpyHostToDevice));
        DS_HIP(h, hipMemcpy(h->chain_buf[G_FIXPREV], s + n13, n14, hipMemcpyHostToDevice));
    } else if (chain_hist_bytes(h)) {
        rc = chain_reserve(h, 1); if (rc) return rc;
        DS_HIP(h, hipMemcpy(h->chain_buf[6], s, chain_hist_bytes(h), hipMemcpyHostToDevice));
    }
    return DS_OK;
}

}  // extern "C"
