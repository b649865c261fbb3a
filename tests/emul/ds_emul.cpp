// ds_emul.cpp — TEST INFRASTRUCTURE: serial CPU execution of the *kernel block program*
// (distantspeech_amd/csrc/ds_core.hpp) so that its index arithmetic and recursions can be
// unit-tested in the CPU-only container.  Built by tests/emul/build.py into tests/emul/libds_emul.so,
// loaded only by tests/test_kernel_emul.py.  The product package never loads it and has no CPU path.
#include <cstring>
#include <vector>

// op_mcspp's direct eigen-solve reports its Laguerre step count here (emul_laguerre_stats)
static int g_lag_it = 0; static int g_laguerre_max = 0; static long long g_laguerre_sum = 0, g_laguerre_n = 0, g_laguerre_hist[32];
static int* lag_note() {            // the previous solve's count goes into the statistics when the next one asks for the counter (and in emul_laguerre_stats)
    if (g_lag_it >= 0) { g_laguerre_max = g_lag_it > g_laguerre_max ? g_lag_it : g_laguerre_max; g_laguerre_sum += g_lag_it; g_laguerre_n += 1; g_laguerre_hist[g_lag_it & 31] += 1; }
    g_lag_it = 0;
    return &g_lag_it;
}
#define DS_LAGUERRE_COUNT lag_note()
#include "../../distantspeech_amd/csrc/ds_core.hpp"
#include "../../distantspeech_amd/csrc/ds_pipe.hpp"
#include "../../distantspeech_amd/csrc/ds_ops.hpp"
#include "../../distantspeech_amd/csrc/ds_quad.hpp"
#include "../../distantspeech_amd/csrc/ds_tables.hpp"
#include "../../distantspeech_amd/csrc/ds_tdfilter.hpp"
#include "../../distantspeech_amd/csrc/ds_fdaf.hpp"
#include "../../distantspeech_amd/csrc/ds_wpe_wide.hpp"
#include "../../distantspeech_amd/csrc/ds_wpe2.hpp"
#include "../../distantspeech_amd/csrc/ds_wpe64.hpp"

namespace {

template <class Rg> struct CpuExec {
    std::vector<Rg> R;
    int nt;
    template <class F> void phase(F f) {
        for (int t = 0; t < nt; ++t) f(t, R[t]);
    }
    template <class F> void phase_wave(F f) { phase(f); }       // wave-local hand-off: the same thing when run serially
    template <class F> void stage(F f) { phase(f); }
    template <class F> void stage_wave(F f) { phase(f); }
    // two-part phases (in-place transform stages of the hop-pipelined engine): every thread's loads before any thread's stores
    template <class FL, class FR> void phase_wave2(FL fl, FR fr) { phase(fl); phase(fr); }
    template <class FL, class FR> void phase2(FL fl, FR fr) { phase(fl); phase(fr); }
    void sync() {}
    void lds_load16(void* lds_piece, int lane, const void* src) { std::memcpy((char*)lds_piece + 16 * lane, src, 16); }   // HipExec: global_load_lds
    void lds_load_wait() {}
};

// The Python side of the emulator keeps operator state as logical planes [B][NF][KP]; the kernels keep it as float4 planes
// [B][ceil(NF / 4)][KP][4] (ds_ops.hpp st_index).  StPlanes packs on entry and unpacks on exit.
struct StPlanes {
    float* logical; int B, NF, KP; std::vector<float> phys;
    StPlanes(float* st, int B_, int NF_, int KP_) : logical(st), B(B_), NF(NF_), KP(KP_) {
        if (!st) return;
        phys.assign((size_t)B * ds::st_floats_per_bin(NF) * KP, 0.0f);
        for (int b = 0; b < B; ++b) for (int f = 0; f < NF; ++f) for (int k = 0; k < KP; ++k)
            phys[(size_t)ds::st_index(b, f, k, NF, KP)] = st[((size_t)b * NF + f) * KP + k];
    }
    float* data() { return logical ? phys.data() : nullptr; }
    ~StPlanes() {
        if (!logical) return;
        for (int b = 0; b < B; ++b) for (int f = 0; f < NF; ++f) for (int k = 0; k < KP; ++k)
            logical[((size_t)b * NF + f) * KP + k] = phys[(size_t)ds::st_index(b, f, k, NF, KP)];
    }
};

static int g_pipe_runs = 0;  // calls served by ds::PipeEngine (the test checks that the switch took effect)
static int g_pipe = 0;       // emul_set_pipe(1): 512-point frame programs run as ds::PipeEngine (ds_pipe.hpp) instead of ds::Engine


template <class E> int run_engine_blob(ds::Params p, int batch, int nfft);
template <int NFFT, int M, int ALGO, bool RYY> int run_t(ds::Params p, int batch) {
    if constexpr (NFFT == 512 && ALGO != ds::ALGO_AIC && ALGO != ds::ALGO_ADAPTIVE_PF) {
        if (g_pipe) { ++g_pipe_runs; return run_engine_blob<ds::PipeEngine<NFFT, M, ALGO, RYY>>(p, batch, NFFT); }
    }
    typedef ds::Engine<NFFT, M, ALGO, RYY> E;
    std::vector<float> blob;
    ds::make_table_blob(NFFT, NFFT / 2, blob, p.out_scale);
    std::vector<ds::vec4> blob4(blob.size() / 4);
    std::memcpy((void*)blob4.data(), blob.data(), blob.size() * sizeof(float));
    p.tables = blob4.data();
    typename E::Sh* sh = new typename E::Sh();
    for (int b = 0; b < batch; ++b) {
        CpuExec<typename E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        std::memset((void*)ex.R.data(), 0, sizeof(typename E::Rg) * E::NT);
        E::run(ex, p, b, *sh);
    }
    delete sh;
    return 0;
}

template <class E> int run_engine_blob(ds::Params p, int batch, int nfft) {
    std::vector<float> blob;
    ds::make_table_blob(nfft, nfft / 2, blob, p.out_scale);
    std::vector<ds::vec4> blob4(blob.size() / 4);
    std::memcpy((void*)blob4.data(), blob.data(), blob.size() * sizeof(float));
    p.tables = blob4.data();
    typename E::Sh* sh = new typename E::Sh();
    for (int b = 0; b < batch; ++b) {
        CpuExec<typename E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        std::memset((void*)ex.R.data(), 0, sizeof(typename E::Rg) * E::NT);
        E::run(ex, p, b, *sh);
    }
    delete sh;
    return 0;
}

template <int NFFT, int M> int run_nm(int algo, int ryy, const ds::Params& p, int batch) {
    if (algo == ds::ALGO_FIXED) return run_t<NFFT, M, ds::ALGO_FIXED, false>(p, batch);
    if (algo == ds::ALGO_ADAPTIVE && !ryy) return run_t<NFFT, M, ds::ALGO_ADAPTIVE, false>(p, batch);
    if (algo == ds::ALGO_ADAPTIVE && ryy) return run_t<NFFT, M, ds::ALGO_ADAPTIVE, true>(p, batch);
    if (algo == ds::ALGO_GSC) return run_t<NFFT, M, ds::ALGO_GSC, false>(p, batch);
    if (algo == ds::ALGO_AIC) return run_t<NFFT, M, ds::ALGO_AIC, false>(p, batch);
    if (algo == ds::ALGO_ADAPTIVE_PF) return run_t<NFFT, M, ds::ALGO_ADAPTIVE_PF, false>(p, batch);
    return -1;
}

template <int NFFT> int run_n(int M, int algo, int ryy, const ds::Params& p, int batch) {
    if constexpr (NFFT == 512) {                 // odd channel counts: the 512-point programs only (build time of this library)
        if (M == 3) return run_nm<NFFT, 3>(algo, ryy, p, batch);
        if (M == 5) return run_nm<NFFT, 5>(algo, ryy, p, batch);
    }
    switch (M) {
        case 2: return run_nm<NFFT, 2>(algo, ryy, p, batch);
        case 4: return run_nm<NFFT, 4>(algo, ryy, p, batch);
        case 6: return run_nm<NFFT, 6>(algo, ryy, p, batch);
        case 8: return run_nm<NFFT, 8>(algo, ryy, p, batch);
    }
    return -1;
}

template <class E> int run_engine(ds::Params p, int batch, int nfft, int hop = 0) {
    std::vector<float> blob;
    ds::make_table_blob(nfft, hop ? hop : nfft / 2, blob, p.out_scale);
    std::vector<ds::vec4> blob4(blob.size() / 4);
    std::memcpy((void*)blob4.data(), blob.data(), blob.size() * sizeof(float));
    p.tables = blob4.data();
    typename E::Sh* sh = new typename E::Sh();
    for (int b = 0; b < batch; ++b) {
        CpuExec<typename E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        E::run(ex, p, b, *sh);
    }
    delete sh;
    return 0;
}

template <int NFFT, int CMAX> int run_fdaf(ds::FdafParams p) {
    typedef ds::FdafEngine<NFFT, CMAX> E;
    std::vector<float> blob;
    float os;
    ds::make_table_blob(NFFT, NFFT / 2, blob, os);
    std::vector<ds::vec4> blob4(blob.size() / 4);
    std::memcpy((void*)blob4.data(), blob.data(), blob.size() * sizeof(float));
    p.tables = blob4.data();
    typename E::Sh* sh = new typename E::Sh();
    for (int b = 0; b < p.B; ++b) {
        CpuExec<typename E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        E::run(ex, p, b, *sh);
    }
    delete sh;
    return 0;
}

// the wide-tap program (ds_wpe_wide.hpp): one 64-lane block per (utterance, bin)
template <int CNP, int NCH, int CT = 0, int NTAPS = 0> int run_wpe_wide(const ds::WpeParams& p) {
    typedef ds::WpeWideEngine<CNP, NCH, CT, NTAPS> E;
    typename E::Sh* sh = new typename E::Sh();
    const long long blocks = (long long)p.B * p.K;
    for (long long b = 0; b < blocks; ++b) {
        CpuExec<typename E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        E::run(ex, p, (int)b, *sh);
    }
    delete sh;
    return 0;
}

template <int CT, int NTAPS> int run_wpe2(const ds::WpeParams& p) {          // two rows of P per lane (ds_wpe2.hpp)
    typedef ds::WpeEngine2<CT, NTAPS> E;
    typename E::Sh* sh = new typename E::Sh();
    const int blocks = (int)(((long long)p.B * p.K + E::BPW - 1) / E::BPW);
    for (int b = 0; b < blocks; ++b) {
        CpuExec<typename E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        E::run(ex, p, b, *sh);
    }
    delete sh;
    return 0;
}

template <int LPB, int CT = 0, int NTAPS = 0> int run_wpe(const ds::WpeParams& p) {
    typedef ds::WpeEngine<LPB, CT, NTAPS> E;
    typename E::Sh* sh = new typename E::Sh();
    const int blocks = (int)(((long long)p.B * p.K + E::BPW - 1) / E::BPW);
    for (int b = 0; b < blocks; ++b) {
        CpuExec<typename E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        E::run(ex, p, b, *sh);
    }
    delete sh;
    return 0;
}

// single-channel transforms at nfft 512 / 1024 run one row per wavefront in the product (launch_transform_stft / _istft): same route here
template <class E> int run_rows(ds::Params p, int rows, int nfft) {
    p.rows = rows;
    return run_engine<E>(p, (rows + 3) / 4, nfft);
}
template <int NFFT> int run_tf(int M, bool inverse, const ds::Params& p, int batch) {
    if constexpr (NFFT >= 512) {
        if (M == 1 && (!inverse || p.method == 1))
            return inverse ? run_rows<ds::IstftRowsEngine<NFFT>>(p, batch, NFFT) : run_rows<ds::StftRowsEngine<NFFT>>(p, batch, NFFT);
    }
#define TF(M_) if (M == M_) return inverse ? run_engine<ds::IstftEngine<NFFT, M_>>(p, batch, NFFT) : run_engine<ds::StftEngine<NFFT, M_>>(p, batch, NFFT);
    TF(1) TF(2) TF(3) TF(4) TF(5) TF(6) TF(7) TF(8)
#undef TF
    return -1;
}
// Transform(n_fft, hop_length = n_fft / 4): the quarter-hop engines
template <int NFFT> int run_tf4(int M, bool inverse, const ds::Params& p, int batch) {
#define TF(M_) if (M == M_) return inverse ? run_engine<ds::IstftEngine<NFFT, M_, 4>>(p, batch, NFFT, NFFT / 4) : run_engine<ds::StftEngine<NFFT, M_, false, 4>>(p, batch, NFFT, NFFT / 4);
    TF(1) TF(2) TF(4) TF(5)
#undef TF
    return -1;
}

}  // namespace

// the F blocking filters of an utterance: per-instance operators (fan_form = 0) or the fan form (one (utterance, bin) at a time)
template <int F> static void fan_rows(int op, const ds::OpParams& p) {
    for (int u = 0; u < p.B / F; ++u)
        for (int k = 0; k < p.K; ++k) {
            if (op == ds::OP_SUBRLS) ds::op_subrls_fan<2, F>(p, u, k);
            else ds::op_sublms_fan<2, F>(p, u, k);
        }
}

template <int NFFT> static int run_front(int M, const ds::Params& p, int batch) {
    if (M == 4 && ds::StftEngine<NFFT, 4, true, 2, true>::front_fits(p.fe_L)) return run_engine<ds::StftEngine<NFFT, 4, true, 2, true>>(p, batch, NFFT);
    if (M == 6 && ds::StftEngine<NFFT, 6, true, 2, true>::front_fits(p.fe_L)) return run_engine<ds::StftEngine<NFFT, 6, true, 2, true>>(p, batch, NFFT);
    return -1;
}

template <int NFFT> static int run_stft_cdr(int M, const ds::Params& p, int batch) {
    if (M == 4) return run_engine<ds::StftEngine<NFFT, 4, true>>(p, batch, NFFT);
    if (M == 6) return run_engine<ds::StftEngine<NFFT, 6, true>>(p, batch, NFFT);
    if (M == 8) return run_engine<ds::StftEngine<NFFT, 8, true>>(p, batch, NFFT);
    return -1;
}

// principal eigenvector of one Hermitian M x M matrix (complex128 [M][M], row-major): method 0 = cyclic Jacobi (herm_principal_d), 1 = the direct
// solve (herm_principal_direct_d); it = Laguerre iterations are not exported
template <int M> static void principal_one(int method, const double* A, double* v) {
    ds::cd Am[M][M], vv[M];
    for (int i = 0; i < M; ++i) for (int j = 0; j < M; ++j) Am[i][j] = ds::mkd(A[2 * (i * M + j)], A[2 * (i * M + j) + 1]);
    int it = 0;
    if (method == 0) ds::herm_principal_d<M>(Am, vv); else ds::herm_principal_direct_d<M>(Am, vv, &it);
    g_laguerre_max = it > g_laguerre_max ? it : g_laguerre_max; g_laguerre_sum += it; g_laguerre_n += 1;
    for (int i = 0; i < M; ++i) { v[2 * i] = vv[i].x; v[2 * i + 1] = vv[i].y; }
}

extern "C" {

// Transform.stft: x -> Y complex [B][T][K][M]; tail_in [B][M][hop] carried
int emul_stft_ov(int nfft, int ov, int M, int batch, const float* x, int layout, int n_samples, float* Y, float* tail_in) {
    ds::Params p;
    std::memset(&p, 0, sizeof p);
    const int hop = nfft / ov, K = nfft / 2 + 1, T = n_samples / hop;
    p.x = x; p.y = Y;
    p.x_batch_stride = (long long)M * n_samples;
    p.y_batch_stride = (long long)T * K * M * 2;
    if (layout == 1) { p.x_sample_stride = 1; p.x_chan_stride = n_samples; } else { p.x_sample_stride = M; p.x_chan_stride = 1; }
    p.T = T; p.tail_in = tail_in;
    if (ov == 4) switch (nfft) { case 256: return run_tf4<256>(M, false, p, batch); case 512: return run_tf4<512>(M, false, p, batch); case 1024: return run_tf4<1024>(M, false, p, batch); }
    if (ov == 2) switch (nfft) { case 256: return run_tf<256>(M, false, p, batch); case 512: return run_tf<512>(M, false, p, batch); case 1024: return run_tf<1024>(M, false, p, batch); }
    return -1;
}
int emul_stft(int nfft, int M, int batch, const float* x, int layout, int n_samples, float* Y, float* tail_in) {
    return emul_stft_ov(nfft, 2, M, batch, x, layout, n_samples, Y, tail_in);
}

// Transform.istft: Y complex [B][T][K][C] -> y [B][T*hop][C]; tail_out [B][M][hop] carried
int emul_istft_ov(int nfft, int ov, int M, int batch, const float* Y, int T, int C, float* y, float* tail_out) {
    ds::Params p;
    std::memset(&p, 0, sizeof p);
    const int hop = nfft / ov, K = nfft / 2 + 1;
    p.x = Y; p.y = y;
    p.x_batch_stride = (long long)T * K * C * 2;
    p.y_batch_stride = (long long)T * hop * C;
    p.T = T; p.method = C; p.tail_out = tail_out;
    if (ov == 4) switch (nfft) { case 256: return run_tf4<256>(M, true, p, batch); case 512: return run_tf4<512>(M, true, p, batch); case 1024: return run_tf4<1024>(M, true, p, batch); }
    if (ov == 2) switch (nfft) { case 256: return run_tf<256>(M, true, p, batch); case 512: return run_tf<512>(M, true, p, batch); case 1024: return run_tf<1024>(M, true, p, batch); }
    return -1;
}
int emul_istft(int nfft, int M, int batch, const float* Y, int T, int C, float* y, float* tail_out) {
    return emul_istft_ov(nfft, 2, M, batch, Y, T, C, y, tail_out);
}

// OpParams fields outside emul_op's argument list (set before the call, sticky)
static int g_repeat = 0, g_two_path = 0;
static int g_wpe_generic = 0;     // emul_set_wpe_generic(1): every WPE shape through the run-time-shape program (the A side of an A/B)
void emul_laguerre_hist(long long* h, int reset) { for (int i = 0; i < 32; ++i) { h[i] = g_laguerre_hist[i]; if (reset) g_laguerre_hist[i] = 0; } }
// max steps << 40 | total steps; *n_solves (optional) = solves counted; reset != 0 clears the counters
long long emul_laguerre_stats(int reset, long long* n_solves) { if (g_lag_it >= 0 && g_laguerre_n + g_laguerre_sum + g_lag_it > 0) { lag_note(); } g_lag_it = -1; const long long r = ((long long)g_laguerre_max << 40) | g_laguerre_sum; if (n_solves) *n_solves = g_laguerre_n; if (reset) { g_laguerre_max = 0; g_laguerre_sum = 0; g_laguerre_n = 0; } return r; }
int emul_principal(int M, int method, int n, const double* A, double* v) {
    for (int q = 0; q < n; ++q) {
        const double* a = A + (size_t)q * 2 * M * M; double* o = v + (size_t)q * 2 * M;
        switch (M) {
            case 2: principal_one<2>(method, a, o); break;
            case 3: principal_one<3>(method, a, o); break;
            case 4: principal_one<4>(method, a, o); break;
            case 5: principal_one<5>(method, a, o); break;
            case 6: principal_one<6>(method, a, o); break;
            case 8: principal_one<8>(method, a, o); break;
            default: return -1;
        }
    }
    return 0;
}
void emul_set_repeat(int on) { g_repeat = on; }
// OP_MCSPP_STEADY_FAN: the SubbandRLS instances the fused operator runs beside McSpp — state [B * M][NF][KP] logical planes, reference input
// x complex [B][T][K], errors e complex [B * M][T][K]
static struct { float* st; int NF; const float* x; float* e; float lam, mu; } g_fan = {nullptr, 0, nullptr, nullptr, 0.0f, 0.0f};
void emul_set_fan(float* st, int NF, const float* x, float* e, float lam, float mu) { g_fan.st = st; g_fan.NF = NF; g_fan.x = x; g_fan.e = e; g_fan.lam = lam; g_fan.mu = mu; }
void emul_set_wpe_generic(int on) { g_wpe_generic = on; }
void emul_set_pipe(int on) { g_pipe = on; }
int emul_pipe_runs() { return g_pipe_runs; }
void emul_set_two_path(int on) { g_two_path = on; }

// frame-level operators: serial loop over (b, k) of ds::run_op
int emul_op(int op, int B, int K, int T, float* st, int NF, const float* in0, const float* in1, const float* in2, float* out0,
            float* out1, float* out2, float* out3, float* out4, int M, int N, int frm_cnt, int ell, int L, int first_frame, int in_complex, int has_p,
            int norm, float mu, float alpha, float reg, float lam) {
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    StPlanes planes(st, B, NF, ds::plane_len(K));
    p.B = B; p.K = K; p.KP = ds::plane_len(K); p.T = T; p.st = planes.data(); p.NF = NF;
    p.in0 = in0; p.in1 = in1; p.in2 = in2; p.out0 = out0; p.out1 = out1; p.out2 = out2; p.out3 = out3; p.out4 = out4;
    p.M = M; p.N = N; p.frm_cnt = frm_cnt; p.ell = ell; p.L = L; p.first_frame = first_frame;
    p.in_complex = in_complex; p.has_p = has_p; p.norm = norm; p.mu = mu; p.alpha = alpha; p.reg = reg; p.lam = lam;
    p.x_fan = 1; p.repeat = g_repeat;
    if (op == ds::OP_MCSPP_STEADY_FAN) {                // the blocking filters' side: emul_set_fan() before the call
        if (!g_fan.st) return -2;
        StPlanes fplanes(g_fan.st, B * M, g_fan.NF, ds::plane_len(K));
        ds::OpParams q = p;
        q.st = fplanes.data(); q.NF = g_fan.NF; q.B = B * M;
        p.fan_st = q.st; p.fan_NF = q.NF; p.fan_x = g_fan.x; p.fan_e = g_fan.e; p.fan_lam = g_fan.lam; p.fan_mu = g_fan.mu;
        p.fan_ctx = &q;
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k) ds::run_op(op, p, b, k);
        return 0;
    }
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k) ds::run_op(op, p, b, k);
    return 0;
}


int emul_fan(int op, int fan_form, int F, int B, int K, int T, float* st, int NF, const float* in0, const float* in1, const float* in2,
             float* out0, int has_p, int norm, float mu, float alpha, float reg, float lam) {
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    StPlanes planes(st, B, NF, ds::plane_len(K));
    p.B = B; p.K = K; p.KP = ds::plane_len(K); p.T = T; p.st = planes.data(); p.NF = NF;
    p.in0 = in0; p.in1 = in1; p.in2 = in2; p.out0 = out0;
    p.M = 1; p.N = 2; p.has_p = has_p; p.norm = norm; p.mu = mu; p.alpha = alpha; p.reg = reg; p.lam = lam;
    p.x_fan = F; p.d_interleaved = 1;
    if (!fan_form) {
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k) ds::run_op(op, p, b, k);
        return 0;
    }
    if (!(op == ds::OP_SUBRLS ? ds::subrls_fan_ok(p) : ds::sublms_fan_ok(p))) return -1;
    switch (F) { case 2: fan_rows<2>(op, p); break; case 4: fan_rows<4>(op, p); break; case 6: fan_rows<6>(op, p); break; default: fan_rows<8>(op, p); }
    return 0;
}

int emul_adaptive_frames(int B, int K, int T, int M, float* st, int NF, const float* Z, const float* gain, float* Y, const float* steer,
                         int frm_cnt, int ell, int L, int method, float alpha_v, float gate, float diag) {
    ds::OpParams p;
    std::memset(&p, 0, sizeof p);
    StPlanes planes(st, B, NF, ds::plane_len(K));
    p.B = B; p.K = K; p.KP = ds::plane_len(K); p.T = T; p.st = planes.data(); p.NF = NF; p.in0 = Z; p.in1 = gain; p.out0 = Y; p.M = M;
    p.frm_cnt = frm_cnt; p.ell = ell; p.L = L; p.has_p = gain != nullptr;
    p.steer = reinterpret_cast<const ds::cf*>(steer); p.steer_batch_stride = 0;
    p.method = method; p.alpha_v = alpha_v; p.beta_v = ds::complement_of(alpha_v); p.gate = gate; p.diag = diag; p.diag_floor = ds::pivot_floor(diag);
    if (!ds::op_supported(ds::OP_ADAPTIVE, M)) return -1;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < K; ++k) ds::run_op(ds::OP_ADAPTIVE, p, b, k);
    return 0;
}

int emul_wpe(int B, int K, int T, int C, int N, const float* xd, const float* d, float* err, float* state, float lam) {
    ds::WpeParams p;
    std::memset(&p, 0, sizeof p);
    p.B = B; p.K = K; p.T = T; p.C = C; p.N = N; p.xd = xd; p.d = d; p.err = err; p.state = state;
    p.ustride = (long long)K * ds::wpe_bin_floats(C, N); p.lam = lam;
    if (C * N > ds::WPE_CNMAX) {                                 // wide prediction filters: launch_wpe_wide's dispatch (ds_kernels_wpe.hip)
        if (C * N > ds::WPEW_CNMAX || C > ds::WPE_CMAX) return -1;
        if (!g_wpe_generic) {
            if (C == 4 && N == 20) return run_wpe_wide<80, 2, 4, 20>(p);
            if (C == 8 && N == 10) return run_wpe_wide<80, 2, 8, 10>(p);
        }
        return C * N <= 32 ? run_wpe_wide<32, 1>(p) : C * N <= 64 ? run_wpe_wide<64, 2>(p) : run_wpe_wide<80, 2>(p);
    }
    const int lpb = ds::wpe_lanes_per_bin(C * N);
    if (!g_wpe_generic) {                                        // the compile-time shapes of launch_wpe (ds_kernels_ops.hip)
        if (C == 8 && N == 2) return run_wpe2<8, 2>(p);
        if (C == 4 && N == 2) return run_wpe2<4, 2>(p);
        if (C == 4 && N == 4) return run_wpe2<4, 4>(p);
        if (C == 8 && N == 1) return run_wpe<8, 8, 1>(p);
        if (C == 2 && N == 3) return run_wpe<8, 2, 3>(p);
    }
    return lpb == 4 ? run_wpe<4>(p) : lpb == 8 ? run_wpe<8>(p) : run_wpe<16>(p);
}

// the quad-spread 8-microphone MVDR bin program (ds_quad.hpp) in its CPU policy (the four lanes of a quad side by side) next to the
// one-thread program it restates: packed state st[64] (updated in place when gate != 0), a[8], z[8] complex -> Y complex, both ways
int emul_quad_mvdr(float* st_quad, float* st_ref, const float* a, const float* z, int gate, float alpha, float diag, float* Y_quad, float* Y_ref) {
    using namespace ds;
    cf A[8], Z[8];
    for (int m = 0; m < 8; ++m) { A[m] = mk(a[2 * m], a[2 * m + 1]); Z[m] = mk(z[2 * m], z[2 * m + 1]); }
    const float beta = complement_of(alpha);
    if (gate) herm_rank1<8>(st_ref, st_ref + 8, Z, alpha, beta);
    const cf yr = mvdr_output<8>(st_ref, st_ref + 8, diag, A, Z);
    Y_ref[0] = yr.x; Y_ref[1] = yr.y;
    QuadRows<float> rows[4];
    for (int l = 0; l < 4; ++l) quad_unpack(l, [&](int f) { return st_quad[f]; }, rows[l]);
    QuadRows<Q4> R;
    auto gather = [&](auto pick) { Q4 v; for (int l = 0; l < 4; ++l) v.v[l] = pick(rows[l]); return v; };
    R.d0 = gather([](const QuadRows<float>& r) { return r.d0; });
    R.d1 = gather([](const QuadRows<float>& r) { return r.d1; });
    for (int k = 0; k < 3; ++k) { R.r0[k].x = gather([k](const QuadRows<float>& r) { return r.r0[k].x; }); R.r0[k].y = gather([k](const QuadRows<float>& r) { return r.r0[k].y; }); }
    for (int k = 0; k < 7; ++k) { R.r1[k].x = gather([k](const QuadRows<float>& r) { return r.r1[k].x; }); R.r1[k].y = gather([k](const QuadRows<float>& r) { return r.r1[k].y; }); }
    cq<Q4> Aq[8], Zq[8];
    for (int m = 0; m < 8; ++m) { Aq[m] = qmk<Q4>(QuadCpu::splat(A[m].x), QuadCpu::splat(A[m].y)); Zq[m] = qmk<Q4>(QuadCpu::splat(Z[m].x), QuadCpu::splat(Z[m].y)); }
    QuadCpu q;
    if (gate) quad_rank1(q, R, Zq, alpha, beta);
    const cq<Q4> y = quad_mvdr_output(q, R, diag, Aq, Zq);
    for (int l = 0; l < 4; ++l) { Y_quad[2 * l] = y.x.v[l]; Y_quad[2 * l + 1] = y.y.v[l]; }
    for (int l = 0; l < 4; ++l) {
        QuadRows<float> o;
        o.d0 = R.d0.v[l]; o.d1 = R.d1.v[l];
        for (int k = 0; k < 3; ++k) o.r0[k] = qmk<float>(R.r0[k].x.v[l], R.r0[k].y.v[l]);
        for (int k = 0; k < 7; ++k) o.r1[k] = qmk<float>(R.r1[k].x.v[l], R.r1[k].y.v[l]);
        quad_pack(l, o, [&](int f, float v) { st_quad[f] = v; });
    }
    return 0;
}

// sizes of the per-bin state storage for (algo, M, ryy): returns NF (floats per bin; NF KP floats per utterance: NF / 4 float4 planes
// [KP], then NF % 4 floats per bin as a narrow plane, the utterance stride rounded up to 32 floats — ds_core.hpp StateLayout), writes KP
int emul_layout(int algo, int nfft, int M, int ryy, int* kp) {
    const int K = nfft / 2 + 1;
    *kp = ds::plane_len(K);
    int nf = 0;
    if (algo == ds::ALGO_ADAPTIVE) nf = M * M + 5 + (ryy ? M * M : 0);
    if (algo == ds::ALGO_GSC) nf = M * (M + 1) + 2 * (M - 1);
    if (algo == ds::ALGO_ADAPTIVE_PF) nf = M * M + 5 + M * (M + 1);
    return nf;
}

// Params::ref_pow for the next emul_run calls (GSC: the powers GSC.py:281-283 hands to omlsa_multi.estimation), [B][T][K][M]; null = off
static float* g_ref_pow = nullptr;
void emul_set_ref_pow(float* ptr) { g_ref_pow = ptr; }

int emul_run(int algo, int nfft, int M, int ryy, int batch, const float* x, int layout, int n_samples, float* y,
             float* bins, float* tail_in, float* tail_out, int* counters, const float* steer, int steer_per_utt,
             int method, int mcra_L, float alpha_y, float alpha_v, float diag, float gate, float mu) {
    ds::Params p;
    std::memset(&p, 0, sizeof p);
    const int hop = nfft / 2, K = nfft / 2 + 1;
    p.x = x;
    p.y = y;
    p.x_batch_stride = (long long)M * n_samples;
    p.y_batch_stride = n_samples;
    if (layout == 1) { p.x_sample_stride = 1; p.x_chan_stride = n_samples; }
    else { p.x_sample_stride = M; p.x_chan_stride = 1; }
    p.T = n_samples / hop;
    p.batch0 = 0;
    p.bins = reinterpret_cast<ds::vec4*>(bins);
    p.tail_in = tail_in;
    p.tail_out = tail_out;
    p.counters = counters;
    p.steer = reinterpret_cast<const ds::cf*>(steer);
    p.steer_batch_stride = steer_per_utt ? (long long)K * M : 0;
    p.method = method;
    p.mcra_L = mcra_L;
    p.alpha_y = alpha_y; p.alpha_v = alpha_v; p.beta_y = ds::complement_of(alpha_y); p.beta_v = ds::complement_of(alpha_v); p.diag = diag; p.diag_floor = ds::pivot_floor(diag); p.gate = gate; p.mu = mu;
    p.ref_pow = g_ref_pow;
    switch (nfft) {
        case 256: return run_n<256>(M, algo, ryy, p, batch);
        case 512: return run_n<512>(M, algo, ryy, p, batch);
        case 1024: return run_n<1024>(M, algo, ryy, p, batch);
    }
    return -1;
}

// the analysis of the SubbandGSC chain's front end with McCDR as its per-bin program (StftEngine<.., CDR = true>): x [B][M][n] -> Y complex
// [B][T][K][M], Gamma [B][T][K], band mean of 1 - Gamma [B][T]; st = the McSpp stage's planes [B][NF][KP] (rows 0..8 McCDR's)
int emul_stft_cdr(int nfft, int M, int batch, const float* x, int n_samples, float* Y, float* tail_in, float* st, int NF, int frm, int ell,
                  const float* Fn, float* gamma, float* qavg) {
    ds::Params p;
    std::memset(&p, 0, sizeof p);
    const int hop = nfft / 2, K = nfft / 2 + 1;
    p.x = x; p.y = Y;
    p.x_batch_stride = (long long)M * n_samples; p.x_sample_stride = 1; p.x_chan_stride = n_samples;
    p.T = n_samples / hop;
    p.y_batch_stride = (long long)p.T * K * M * 2;
    p.tail_in = tail_in;
    StPlanes planes(st, batch, NF, ds::plane_len(nfft / 2 + 1));
    p.cdr_st = planes.data(); p.cdr_NF = NF; p.cdr_frm = frm; p.cdr_ell = ell; p.cdr_L = 65; p.cdr_fn = Fn; p.cdr_gamma = gamma; p.cdr_qavg = qavg;
    switch (nfft) {
        case 256: return run_stft_cdr<256>(M, p, batch);
        case 512: return run_stft_cdr<512>(M, p, batch);
        case 1024: return run_stft_cdr<1024>(M, p, batch);
    }
    return -1;
}

// the chain's whole front end as one program (StftEngine<.., FRONT = true>): raw x [B][M][n] -> DC notch -> FIR bank + channel mean -> analysis
// + McCDR; notch memories mem [B][M][2], FIR history cache_in -> cache_out [B][M][L - 1], fixed [B][n]
int emul_front(int nfft, int M, int batch, const float* x, int n_samples, float* Y, float* tail_in, float* st, int NF, int frm, int ell,
               const float* Fn, float* gamma, float* qavg, const float* coef, int L, double* mem, const float* cache_in, float* cache_out,
               float* fixed, float radius) {
    ds::Params p;
    std::memset(&p, 0, sizeof p);
    const int hop = nfft / 2, K = nfft / 2 + 1;
    p.x = x; p.y = Y;
    p.x_batch_stride = (long long)M * n_samples; p.x_sample_stride = 1; p.x_chan_stride = n_samples;
    p.T = n_samples / hop;
    p.y_batch_stride = (long long)p.T * K * M * 2;
    p.tail_in = tail_in;
    StPlanes planes(st, batch, NF, ds::plane_len(nfft / 2 + 1));
    p.cdr_st = planes.data(); p.cdr_NF = NF; p.cdr_frm = frm; p.cdr_ell = ell; p.cdr_L = 65; p.cdr_fn = Fn; p.cdr_gamma = gamma; p.cdr_qavg = qavg;
    p.fe_coef = coef; p.fe_L = L; p.fe_mem = mem; p.fe_cache_in = cache_in; p.fe_cache_out = cache_out; p.fe_fixed = fixed; p.fe_radius = ds::decimal_double(radius);
    switch (nfft) {
        case 512: return run_front<512>(M, p, batch);
        case 1024: return run_front<1024>(M, p, batch);
    }
    return -1;
}

// the SubbandGSC chain's tail as one frame program (Engine<.., ALGO_AIC>): blocking-matrix outputs x [B][M][n] -> y [B][n]; st = the
// canceller's planes [B][NF][KP] (the DS_ALGO_SUBLMS operator's layout), d [B][T][K] complex taken one frame late through dprev [B][K]
int emul_aic(int nfft, int M, int batch, const float* x, int n_samples, float* y, float* tail_in, float* tail_out, int* counters,
             float* st, int NF, const float* d, float* dprev, const float* pk, int pc, int norm, float mu, float alpha, float reg,
             const float* e_spectra, float* bmtail, float* bm_out) {
    ds::Params p;
    std::memset(&p, 0, sizeof p);
    const int hop = nfft / 2;
    p.x = x; p.y = y;
    p.x_batch_stride = (long long)M * n_samples; p.y_batch_stride = n_samples;
    p.x_sample_stride = 1; p.x_chan_stride = n_samples;
    p.T = n_samples / hop;
    p.tail_in = tail_in; p.tail_out = tail_out; p.counters = counters;
    p.mcra_L = 1;
    StPlanes planes(st, batch, NF, ds::plane_len(nfft / 2 + 1));
    p.aic_st = planes.data(); p.aic_NF = NF; p.aic_d = d; p.aic_dprev = dprev; p.aic_p = pk; p.aic_pc = pc; p.aic_norm = norm;
    p.aic_mu = mu; p.aic_alpha = alpha; p.aic_reg = reg;
    p.aic_e = e_spectra; p.aic_bmtail = bmtail; p.aic_bm = bm_out;     // e_spectra set: the blocking-matrix synthesis runs inside the program
    switch (nfft) {
        case 256: return run_n<256>(M, ds::ALGO_AIC, 0, p, batch);
        case 512: return run_n<512>(M, ds::ALGO_AIC, 0, p, batch);
        case 1024: return run_n<1024>(M, ds::ALGO_AIC, 0, p, batch);
    }
    return -1;
}

// time-domain front-end: serial loops over the same per-thread programs the GPU runs
int emul_dcnotch(int B, int M, int n, const float* x, float* y, double* mem, float radius) {
    ds::TdParams p;
    std::memset(&p, 0, sizeof p);
    p.B = B; p.M = M; p.n = n; p.x = x; p.y = y; p.mem = mem; p.radius = ds::decimal_double(radius);
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < M; ++m) ds::td_dcnotch(p, b, m);
    return 0;
}

int emul_firbank(int B, int M, int n, int L, const float* x, float* y, float* mean, const float* coef, const float* cache_in,
                 float* cache_out) {
    ds::TdParams p;
    std::memset(&p, 0, sizeof p);
    p.B = B; p.M = M; p.n = n; p.L = L; p.x = x; p.y = y; p.mean = mean; p.coef = coef; p.cache_in = cache_in; p.cache_out = cache_out;
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < n; ++i) ds::td_fir(p, b, i);
        for (int i = 0; i < L - 1; ++i) ds::td_fir_cache(p, b, i);
    }
    return 0;
}

int emul_tdfilter(int mode, int B, int n, int L, const float* x, const float* d, float* err, float* w, float* buf, float* P,
                  float mu, float eps, float pp, float lam, int norm) {
    ds::TdfParams p;
    std::memset(&p, 0, sizeof p);
    p.B = B; p.n = n; p.L = L; p.mode = mode; p.x = x; p.d = d; p.err = err; p.w = w; p.buf = buf; p.P = P;
    p.mu = mu; p.eps = eps; p.p = pp; p.lam = lam; p.norm = norm;
    ds::TdfShared* sh = new ds::TdfShared();
    for (int b = 0; b < B; ++b) {
        CpuExec<ds::TdfRegs> ex;
        ex.nt = ds::TDF_NT;
        ex.R.resize(ds::TDF_NT);
        ds::TdfEngine::run(ex, p, b, *sh);
    }
    delete sh;
    return 0;
}
int emul_fdaf(int nfft, int B, int T, int C, int kind, int constrain, int non_causal, int weight_norm, int trunc, int p_mode,
              float mu, float alpha, const float* x, const float* d, const float* pp, float* err, float* w_out,
              float* state) {
    ds::FdafParams p;
    std::memset(&p, 0, sizeof p);
    p.B = B; p.T = T; p.C = C; p.kind = kind; p.constrain = constrain; p.non_causal = non_causal; p.weight_norm = weight_norm;
    p.trunc = trunc; p.p_mode = p_mode; p.mu = mu; p.alpha = alpha; p.x = x; p.d = d; p.p = pp; p.err = err; p.w_out = w_out;
    p.state = state; p.state_stride = ds::fdaf_state_floats(nfft, C); p.two_path = g_two_path && kind == 0;
#define FD(N_) if (nfft == N_) { if (C == 1 && !p.two_path) return run_fdaf<N_, 1>(p); if (C <= 4) return run_fdaf<N_, 4>(p); if (C <= 8) return run_fdaf<N_, 8>(p); }
    FD(128) FD(256) FD(512) FD(1024)
#undef FD
    return -1;
}

// the RLS-WPE recursion in double (ds_wpe64.hpp): state [B][K][wpe64_bin_doubles] doubles, inputs / outputs complex64 as the fp32 kernels'
int emul_wpe64(int B, int K, int T, int C, int N, const float* xd, const float* d, float* err, double* state, float lam,
               float* ring, int ring_pos, int ring_len) {
    ds::Wpe64Params q;
    std::memset(&q, 0, sizeof q);
    q.w.B = B; q.w.K = K; q.w.T = T; q.w.C = C; q.w.N = N; q.w.xd = xd; q.w.d = d; q.w.err = err; q.w.lam = lam;
    q.w.ring = ring; q.w.ring_pos = ring_pos; q.w.ring_len = ring_len;
    q.state64 = state; q.ustride64 = (long long)K * ds::wpe64_bin_doubles(C, N); q.lam64 = ds::wpe64_lambda(lam);
    const int CN = C * N;
    if (CN < 1 || CN > ds::WPEW_CNMAX || C > ds::WPE_CMAX) return -1;
    typedef ds::Wpe64Engine<80> E;
    E::Sh* sh = new E::Sh();
    for (int g = 0; g < B * K; ++g) {
        CpuExec<E::Rg> ex;
        ex.nt = E::NT;
        ex.R.resize(E::NT);
        E::run(ex, q, g, *sh);
    }
    delete sh;
    return 0;
}
}
