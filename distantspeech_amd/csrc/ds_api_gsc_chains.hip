// ds_api_gsc_chains.hip — the two overlap-save GSCs of the reference as device-resident chain handles:
//   DS_ALGO_TDGSC   TDGSC.process   beamformer/TDGSC.py:110-175
//   DS_ALGO_FDGSC   FDGSC.process   beamformer/FDGSC.py:201-317 (blocking-matrix mode 3)
// Every stage is one of the library's operators working on the previous stage's device buffer on the chain's stream; the block delays
// are device copies, the adaptation-control scalars of FDGSC (mean p, the :248-255 threshold) one small kernel.  Nothing returns to the
// host between the stages.
#include "ds_handle.hpp"

using namespace dsi;

namespace ds {
// FDGSC.py:248-255,282 per (utterance, block): if the mean speech presence probability over bins 32 .. 127 exceeds 0.8, the bins below 32
// are raised to at least 0.8; pa = 1 - mean over all bins of the (modified) row.  One wave per row: every lane adds its bins (stride 64)
// in double, the lanes' sums meet in a fixed butterfly order.
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void fdgsc_control_rows(float* p, float* pa, int rows, int K, int block) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = block * 4 + wv;
    if (r >= rows) return;
    float* row = p + (long long)r * K;
    double mid = 0.0, all = 0.0;
    float first = 0.0f;                                                 // this lane's bin `lane` (the only one the threshold can change)
    for (int k = lane; k < K; k += 64) {
        const float v = row[k];
        if (k == lane) first = v; else all += (double)v;
        if (k >= 32 && k < 128) mid += (double)v;
    }
    mid = wave_sum(mid);
    const int nmid = (K < 128 ? K : 128) - 32;                          // np.mean(p_bm[32:128]) is over the bins that exist (frameLen 64: 33 of them)
    if (nmid > 0 && mid / (double)nmid > 0.8 && lane < 32 && first < 0.8f) { first = 0.8f; row[lane] = 0.8f; }
    if (lane < K) all += (double)first;
    all = wave_sum(all);
    if (lane == 0) pa[r] = (float)(1.0 - all / (double)K);
}
// FDGSC.py:258,270: the two block delays of a call together (round 6; they were six strided device copies) — rows [0, BM): aligned channel
// rows delayed by half a block (delay_aligned, H samples of carried tail), rows [BM, BM + B): the fixed beamformer output delayed by one block
// (delay_fbf, FL samples of carried tail).  A row's carried tail is read into registers before the row's new tail is written (one workgroup per
// row).  tf_tail (optional): Transform.previous_input of the transform_fbf the delayed fixed output is analysed with when no post-filter runs —
// the analysis there only advances that state (FDGSC.py:273), which is the last block of the delayed row: written here instead of launching it.
__device__ __forceinline__ void fdgsc_delay_row(const float* xa, float* xad, float* altail, const float* fixed, float* fixd, float* fixprev,
                                                float* tf_tail, int BM, int N, int H, int FL, int r) {
    const int tid = threadIdx.x;
    const bool al = r < BM;
    const int D = al ? H : FL;                                          // this row's delay
    const float* src = al ? xa + (long long)r * N : fixed + (long long)(r - BM) * N;
    float* dst = al ? xad + (long long)r * N : fixd + (long long)(r - BM) * N;
    float* tail = al ? altail + (long long)r * H : fixprev + (long long)(r - BM) * FL;
    float keep[4];                                                      // D <= 1024 samples: four per lane at most
#pragma unroll
    for (int j = 0; j < 4; ++j) keep[j] = tid + 256 * j < D ? tail[tid + 256 * j] : 0.0f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) if (tid + 256 * j < D) { dst[tid + 256 * j] = keep[j]; tail[tid + 256 * j] = src[N - D + tid + 256 * j]; }
    for (int i = D + tid; i < N; i += 256) dst[i] = src[i - D];
    if (!al && tf_tail) {                                               // the last FL samples of the delayed row
        float* t = tf_tail + (long long)(r - BM) * FL;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int i = tid + 256 * j; if (i < FL) t[i] = N > FL ? src[N - 2 * FL + i] : keep[j]; }
    }
}
// ... and both in ONE launch (they are independent: the control reads the MCRA's p, the delays the FIR bank's outputs): workgroups [0, BM + B)
// take a delay row each, the rest four rows of p each
__global__ void __launch_bounds__(256) ds_fdgsc_prep_kernel(float* p, float* pa, int rows, int K, const float* xa, float* xad, float* altail,
                                                             const float* fixed, float* fixd, float* fixprev, float* tf_tail, int BM, int B, int N,
                                                             int H, int FL) {
    const int blk = blockIdx.x;
    if (blk < BM + B) fdgsc_delay_row(xa, xad, altail, fixed, fixd, fixprev, tf_tail, BM, N, H, FL, blk);
    else fdgsc_control_rows(p, pa, rows, K, blk - (BM + B));
}
}  // namespace ds

namespace dsi {

// buffer slots of the two chains (indices into ds_handle::chain_buf; "c" = complex64)
enum {
    Q_XN = 0,       // [B][M][n]         input after the DC notch
    Q_XA = 1,       // TDGSC [B][n][M] / FDGSC [B][M][n]   time-aligned channels
    Q_FIXED = 2,    // [B][n]            fixed beamformer output (channel mean)
    Q_BM = 3,       // TDGSC [B][n][M-1] pairwise differences / FDGSC [B][M][n] blocking-filter outputs
    Q_D = 4,        // c[B][T][K]        analysis of the MCRA input
    Q_LAM = 5,      // [B][T][K]         MCRA noise estimate (not returned)
    Q_P = 6,        // [B][T][K]         speech presence probability
    Q_OUT = 7,      // [B][n]            canceller output
    Q_Y = 8,        // c[B][T][K]        its analysis (post-filter)
    Q_U = 9,        // c[B][T][K][M-1]   reference spectra (post-filter)
    Q_G = 10,       // [B][T][K]         OMLSA gain
    Q_Y2 = 11,      // c[B][T][K]        post-filtered spectrum
    Q_XAD = 12,     // FDGSC [B][M][n]   aligned channels delayed by half a block
    Q_FIXD = 13,    // FDGSC [B][n]      fixed beamformer output delayed by one block
    Q_PA = 14,      // FDGSC [B][T]      1 - mean p per block
    Q_SCR = 15,     // c[B][T][K]        analysis results that only advance a transform's state
    Q_ALTAIL = 16,  // FDGSC state [B][M][hop/2]   delay_aligned (:96)
    Q_FIXPREV = 17, // FDGSC state [B][hop]        delay_fbf (:93)
    Q_BMLAST = 18,  // FDGSC state [B][M][hop]     last block of bm_output of the previous call (post-filter re-analysis, :288-291)
    Q_ZERO = 19     // [B][M][hop] zeros
};

static int reserve(ds_handle* h, int i, size_t bytes, bool zero) {
    if (bytes <= h->chain_bytes[i]) return DS_OK;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->chain_buf[i]); h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0;
    DS_HIP(h, hipMalloc((void**)&h->chain_buf[i], bytes));
    h->chain_bytes[i] = bytes;
    if (zero) DS_HIP(h, hipMemset(h->chain_buf[i], 0, bytes));
    return DS_OK;
}

// sub-handles.  TDGSC (TDGSC.py:24-50): 0 front end (notch radius 0.98 + TimeAlignment), 1 transform (1 ch), 2 MCRA L = 65, 3 canceller
// FastFreqLms(filter_len, n_channels = M - 1, non_causal), 4 transform_fbf (1), 5 transform_bm (M - 1), 6 NsOmlsaMulti(M).
// FDGSC (FDGSC.py:38-116): 0 front end, 1 transform_x channel 0 (1 ch), 2 MCRA L = 60, 3 the M blocking filters as B * M instances of
// AdaptiveBlockingMatrixFilter(mu 0.1, alpha 0.9), 4 AdaptiveInterferenceCancellation(M channels, mu 0.1, alpha 0.9, weight_norm),
// 5 transform_fbf (1), 6 the re-analysis transform of the references (M - 1), 7 NsOmlsaMulti(M).
int gsc_chain_create(ds_handle* h) {
    const ds_config& cfg = h->cfg;
    const int M = cfg.n_mics;
    const bool td = cfg.algo == DS_ALGO_TDGSC;
    const int ns = td ? 7 : 8;
    for (int i = 0; i < ns; ++i) {
        ds_config c = cfg;
        c.device = h->device; c.filter_len = 0; c.filt_mu = 0.0f; c.filt_alpha = 0.0f; c.rls_lambda = 0.0f; c.mcra_L = 0;
        int kind = -1, non_causal = 0, weight_norm = 0;
        if (i == 0) { c.algo = DS_ALGO_FRONTEND; c.filt_alpha = 0.98f; }
        else if (i == 1) { c.algo = DS_ALGO_TRANSFORM; c.n_mics = 1; }
        else if (i == 2) { c.algo = DS_ALGO_MCRA; c.n_mics = 1; c.mcra_L = td ? 65 : 60; }
        else if (td) {
            if (i == 3) { c.algo = DS_ALGO_FDAF; c.n_mics = M - 1; c.filt_mu = 0.01f; c.filt_alpha = 0.9f; kind = DS_FDAF_PLAIN; non_causal = 1; }
            else if (i == 4) { c.algo = DS_ALGO_TRANSFORM; c.n_mics = 1; }
            else if (i == 5) { c.algo = DS_ALGO_TRANSFORM; c.n_mics = M - 1; }
            else { c.algo = DS_ALGO_OMLSA; }
        } else {
            if (i == 3) { c.algo = DS_ALGO_FDAF; c.n_mics = 1; c.batch = cfg.batch * M; c.filt_mu = 0.1f; c.filt_alpha = 0.9f; kind = DS_FDAF_BM; }
            else if (i == 4) { c.algo = DS_ALGO_FDAF; c.filt_mu = 0.1f; c.filt_alpha = 0.9f; kind = DS_FDAF_AIC; weight_norm = 1; }
            else if (i == 5) { c.algo = DS_ALGO_TRANSFORM; c.n_mics = 1; }
            else if (i == 6) { c.algo = DS_ALGO_TRANSFORM; c.n_mics = M - 1; }
            else { c.algo = DS_ALGO_OMLSA; }
        }
        int rc = ds_create(&c, &h->sub[i]);
        if (rc != DS_OK) return fail(h, rc, "ds_create(TDGSC / FDGSC chain): stage " + std::to_string(i) + ": " + g_err);
        (void)hipStreamDestroy(h->sub[i]->stream);
        h->sub[i]->stream = h->stream; h->sub[i]->owns_stream = false;
        if (c.algo == DS_ALGO_MCRA) h->sub[i]->mcra_L = c.mcra_L;
        if (kind >= 0) {
            h->sub[i]->fdaf_kind = kind; h->sub[i]->fdaf_constrain = 1; h->sub[i]->fdaf_non_causal = non_causal; h->sub[i]->fdaf_weight_norm = weight_norm;
        }
    }
    return DS_OK;
}

// analysis / synthesis of a transform sub-handle on device buffers with explicit strides
static int tf_stft(ds_handle* h, ds_handle* t, const float* x, long long bstride, long long sstride, long long cstride, int n, float* Y) {
    Params p;
    fill_params(t, p);
    const int C = t->cfg.n_mics, T = n / t->cfg.hop;
    p.x = x; p.y = Y;
    p.x_batch_stride = bstride; p.x_sample_stride = sstride; p.x_chan_stride = cstride;
    p.y_batch_stride = (long long)T * t->K * C * 2;
    p.T = T; p.batch0 = 0;
    DS_HIP(h, launch_transform_stft(t, p, t->cfg.batch, t->stream));
    return DS_OK;
}
static int tf_istft(ds_handle* h, ds_handle* t, const float* Y, int T, float* y, long long y_bstride) {
    Params p;
    fill_params(t, p);
    p.x = Y; p.y = y;
    p.x_batch_stride = (long long)T * t->K * 2;
    p.y_batch_stride = y_bstride;
    p.T = T; p.batch0 = 0; p.method = 1;
    DS_HIP(h, launch_transform_istft(t, p, t->cfg.batch, t->stream));
    return DS_OK;
}

#define GS_SUB(i, call) do { int rc_ = (call); if (rc_) return fail(h, rc_, h->sub[i]->err); } while (0)

static int front_end(ds_handle* h, const float* x, long long x_bstride, long long x_cstride, int n, bool notch, bool xa_chan_major, float* diff) {
    ds_handle* fe = h->sub[0];
    const int B = h->cfg.batch, M = h->cfg.n_mics;
    if (fe->aux_floats == 0 || fe->aux_floats % M != 0) return fail(h, DS_ESTATE, "TDGSC / FDGSC chain: call ds_chain_set_aux(DS_CHAIN_AUX_FIR) first");
    float** cb = h->chain_buf;
    ds::TdParams p;
    const float* fir_in = x;
    long long fin_b = x_bstride, fin_c = x_cstride;
    if (notch) {
        std::memset(&p, 0, sizeof p);
        p.B = B; p.M = M; p.n = n; p.x = x; p.x_bstride = x_bstride; p.x_cstride = x_cstride; p.y = cb[Q_XN]; p.mem = fe->td_mem;
        p.radius = ds::decimal_double(fe->cfg.filt_alpha);
        DS_HIP(h, ds::launch_dcnotch(p, h->stream));
        fir_in = cb[Q_XN]; fin_b = (long long)M * n; fin_c = n;
    } else if (x_bstride != (long long)M * n || x_cstride != n) {
        DS_HIP(h, hipMemcpy2DAsync(cb[Q_XN], (size_t)n * 4, x, (size_t)x_cstride * 4, (size_t)n * 4, (size_t)B * M, hipMemcpyDeviceToDevice, h->stream));
        fir_in = cb[Q_XN];
    }
    (void)fin_b; (void)fin_c;
    const int Lt = (int)(fe->aux_floats / M);
    int rc = frontend_set_taps(fe, Lt); if (rc) return fail(h, rc, fe->err);
    std::memset(&p, 0, sizeof p);
    p.B = B; p.M = M; p.n = n; p.L = Lt; p.x = fir_in; p.x_chan_major = 1; p.y = cb[Q_XA]; p.y_chan_major = xa_chan_major ? 1 : 0; p.mean = cb[Q_FIXED];
    p.diff = diff;
    p.coef = fe->dev_buf[9]; p.cache_in = fe->td_cache[fe->td_cur]; p.cache_out = fe->td_cache[fe->td_cur ^ 1];
    DS_HIP(h, ds::launch_fir(p, h->stream));
    fe->td_cur ^= 1;
    return DS_OK;
}

// OMLSA post-filter of a signal of T blocks against reference spectra U (TDGSC.py:158-170 / FDGSC.py:286-298): analysis and synthesis on
// the shared transform_fbf `tf`; U_frames = T (one reference frame per block) or 1 (the same frame for every block)
static int postfilter_blocks(ds_handle* h, ds_handle* tf, ds_handle* om, const float* sig, long long sig_bstride, int n, const float* U, float* out,
                             long long out_bstride) {
    float** cb = h->chain_buf;
    const int T = n / h->cfg.hop;
    int rc = tf_stft(h, tf, sig, sig_bstride, 1, sig_bstride, n, cb[Q_Y]); if (rc) return rc;
    rc = ds_omlsa_postfilter(om, cb[Q_Y], U, T, cb[Q_G], cb[Q_Y2], DS_MEM_DEVICE); if (rc) return fail(h, rc, om->err);
    return tf_istft(h, tf, cb[Q_Y2], T, out, out_bstride);
}

int tdgsc_run(ds_handle* h, const float* x, long long x_bstride, long long x_cstride, int n, int postfilter, float* out, long long out_bstride,
              float* p_dev, float* bm_dev, float* w_dev) {
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, K = h->K, T = n / h->cfg.hop, N = n;
    const size_t need[Q_SCR + 1] = {B * M * N * 4, B * N * M * 4, B * N * 4, B * N * (M - 1) * 4, B * T * K * 8, B * T * K * 4, B * T * K * 4, B * N * 4,
                                    postfilter ? B * T * K * 8 : 0, postfilter ? B * T * K * (M - 1) * 8 : 0, postfilter ? B * T * K * 4 : 0,
                                    postfilter ? B * T * K * 8 : 0, 0, 0, 0, 0};
    for (int i = 0; i <= Q_SCR; ++i) if (need[i]) { rc = reserve(h, i, need[i], false); if (rc) return rc; }
    float** cb = h->chain_buf;
    rc = front_end(h, x, x_bstride, x_cstride, n, true, false, cb[Q_BM]); if (rc) return rc;                 // :129-130,143,149
    rc = tf_stft(h, h->sub[1], cb[Q_FIXED], n, 1, n, n, cb[Q_D]); if (rc) return rc;                          // :145
    GS_SUB(2, ds_mcra_estimate_p(h->sub[2], cb[Q_D], 1, (int)T, cb[Q_LAM], cb[Q_P], DS_MEM_DEVICE));           // :146-147
    float* o1 = postfilter ? cb[Q_OUT] : out;
    if (!postfilter && out_bstride != (long long)N) o1 = cb[Q_OUT];
    // :152-156 -> :105: the canceller adapts with p = 1 - p, coefficients truncated to [30, filter_len - 30)
    GS_SUB(3, fdaf_run_dev(h->sub[3], cb[Q_BM], cb[Q_FIXED], cb[Q_P], DS_FDAF_P_BIN | DS_FDAF_P_COMPLEMENT, (int)T, 30, o1, w_dev, 0, 0, 0, 0));
    if (postfilter) {                                                                                        // :158-170
        rc = tf_stft(h, h->sub[5], cb[Q_BM], (long long)n * (M - 1), (long long)(M - 1), 1, n, cb[Q_U]); if (rc) return rc;
        rc = postfilter_blocks(h, h->sub[4], h->sub[6], cb[Q_OUT], (long long)N, n, cb[Q_U], out, out_bstride); if (rc) return rc;
    } else if (o1 != out) {
        DS_HIP(h, hipMemcpy2DAsync(out, (size_t)out_bstride * 4, o1, N * 4, N * 4, B, hipMemcpyDeviceToDevice, h->stream));
    }
    if (p_dev) DS_HIP(h, hipMemcpyAsync(p_dev, cb[Q_P], B * T * K * 4, hipMemcpyDeviceToDevice, h->stream));
    if (bm_dev) DS_HIP(h, hipMemcpyAsync(bm_dev, cb[Q_BM], B * N * (M - 1) * 4, hipMemcpyDeviceToDevice, h->stream));
    return DS_OK;
}

int fdgsc_run(ds_handle* h, const float* x, long long x_bstride, long long x_cstride, int n, int postfilter, int dc_notch, float* out,
              long long out_bstride, float* p_dev, float* fix_dev, float* fixd_dev, float* bm_dev, float* al_dev, float* ald_dev,
              float* waic_dev, float* wbm_dev) {
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, K = h->K, FL = h->cfg.hop, H = FL / 2, T = n / FL, N = n;
    const size_t need[Q_ZERO + 1] = {B * M * N * 4, B * M * N * 4, B * N * 4, B * M * N * 4, B * T * K * 8, B * T * K * 4, B * T * K * 4, B * N * 4,
                                     postfilter ? B * K * 8 : 0, postfilter ? B * K * (M - 1) * 8 : 0, postfilter ? B * K * 4 : 0,
                                     postfilter ? B * K * 8 : 0, B * M * N * 4, B * N * 4, B * T * 4, B * T * K * 8,
                                     B * M * H * 4, B * FL * 4, B * M * FL * 4, B * M * FL * 4};
    for (int i = 0; i <= Q_ZERO; ++i) if (need[i]) { rc = reserve(h, i, need[i], i >= Q_ALTAIL); if (rc) return rc; }
    float** cb = h->chain_buf;
    rc = front_end(h, x, x_bstride, x_cstride, n, dc_notch != 0, true, nullptr); if (rc) return rc;          // :213-215,235,238
    const float* xin = dc_notch ? cb[Q_XN] : x;                                                              // the MCRA watches channel 0 of the (notched) input
    rc = tf_stft(h, h->sub[1], xin, dc_notch ? (long long)(M * N) : x_bstride, 1, N, n, cb[Q_D]); if (rc) return rc;   // :241
    GS_SUB(2, ds_mcra_estimate_p(h->sub[2], cb[Q_D], 1, (int)T, cb[Q_LAM], cb[Q_P], DS_MEM_DEVICE));           // :243-244
    // :248-255,282 the adaptation control from the MCRA's p, :258 delay_aligned (half a block, channel-major rows [B * M][n]) and :270 delay_fbf (one
    // block) in one launch; without the post-filter the analysis of the delayed fixed output (:273) only advances transform_fbf's carried
    // input block: that block is written by the same launch
    hipLaunchKernelGGL(ds::ds_fdgsc_prep_kernel, dim3((unsigned)(B * M + B + (B * T + 3) / 4)), dim3(256), 0, h->stream, cb[Q_P], cb[Q_PA], (int)(B * T), (int)K,
                       cb[Q_XA], cb[Q_XAD], cb[Q_ALTAIL], cb[Q_FIXED], cb[Q_FIXD], cb[Q_FIXPREV], postfilter ? nullptr : h->sub[5]->tail_in, (int)(B * M),
                       (int)B, (int)N, (int)H, (int)FL);
    DS_HIP(h, hipGetLastError());
    // :259-264 -> :185-195: the M blocking filters of an utterance share the fixed-beamformer output; desired = delayed aligned channel m
    GS_SUB(3, fdaf_run_dev(h->sub[3], cb[Q_FIXED], cb[Q_XAD], nullptr, DS_FDAF_P_NONE, (int)T, -1, cb[Q_BM], wbm_dev, (int)M, (long long)N, 1, 0));
    // :278-284: canceller input = the blocking-filter outputs (channel-major), desired = delayed fixed output, p = 1 - mean p per block
    float* o1 = (postfilter || out_bstride != (long long)N) ? cb[Q_OUT] : out;
    GS_SUB(4, fdaf_run_dev(h->sub[4], cb[Q_BM], cb[Q_FIXD], cb[Q_PA], DS_FDAF_P_BLOCK, (int)T, -1, o1, waic_dev, 1, (long long)(M * N), 1, (long long)N));
    if (postfilter) {
        // :273,286-298 block by block: transform_fbf serves the delayed fixed output AND the canceller output (frame t of either analysis
        // starts from the other signal's previous block), and the reference re-analyses the WHOLE bm_output array every block and keeps
        // frame 0 = [last hop the reference transform saw | block 0 of this array]: constant from the second block of a call on
        ds_handle *tf = h->sub[5], *tu = h->sub[6], *om = h->sub[7];
        for (size_t blk = 0; blk < T; ++blk) {
            rc = tf_stft(h, tf, cb[Q_FIXD] + blk * FL, (long long)N, 1, (long long)N, (int)FL, cb[Q_SCR]); if (rc) return rc;
            if (blk <= 1) {
                rc = ds_reset(tu); if (rc) return fail(h, rc, tu->err);
                const float* prev = blk == 0 ? cb[Q_BMLAST] : cb[Q_ZERO];
                rc = tf_stft(h, tu, prev, (long long)(M * FL), 1, (long long)FL, (int)FL, cb[Q_U]); if (rc) return rc;
                rc = tf_stft(h, tu, cb[Q_BM], (long long)(M * N), 1, (long long)N, (int)FL, cb[Q_U]); if (rc) return rc;   // channels 0 .. M-2 of block 0
            }
            rc = postfilter_blocks(h, tf, om, cb[Q_OUT] + blk * FL, (long long)N, (int)FL, cb[Q_U], out + blk * FL, out_bstride);
            if (rc) return rc;
        }
        DS_HIP(h, hipMemcpy2DAsync(cb[Q_BMLAST], FL * 4, (char*)cb[Q_BM] + (N - FL) * 4, N * 4, FL * 4, B * M, hipMemcpyDeviceToDevice, h->stream));
    } else {
        // (:273 the analysis of the delayed fixed output keeps advancing transform_fbf's state: done by the delays kernel above)
        if (o1 != out) DS_HIP(h, hipMemcpy2DAsync(out, (size_t)out_bstride * 4, o1, N * 4, N * 4, B, hipMemcpyDeviceToDevice, h->stream));
    }
    if (p_dev) DS_HIP(h, hipMemcpyAsync(p_dev, cb[Q_P], B * T * K * 4, hipMemcpyDeviceToDevice, h->stream));
    if (fix_dev) DS_HIP(h, hipMemcpyAsync(fix_dev, cb[Q_FIXED], B * N * 4, hipMemcpyDeviceToDevice, h->stream));
    if (fixd_dev) DS_HIP(h, hipMemcpyAsync(fixd_dev, cb[Q_FIXD], B * N * 4, hipMemcpyDeviceToDevice, h->stream));
    if (bm_dev) DS_HIP(h, hipMemcpyAsync(bm_dev, cb[Q_BM], B * M * N * 4, hipMemcpyDeviceToDevice, h->stream));
    if (al_dev) DS_HIP(h, hipMemcpyAsync(al_dev, cb[Q_XA], B * M * N * 4, hipMemcpyDeviceToDevice, h->stream));
    if (ald_dev) DS_HIP(h, hipMemcpyAsync(ald_dev, cb[Q_XAD], B * M * N * 4, hipMemcpyDeviceToDevice, h->stream));
    return DS_OK;
}

}  // namespace dsi

extern "C" {

int ds_tdgsc_process(ds_handle* h, const float* x, int n_samples, int postfilter, float* out, float* p, float* bm, float* w, int mem) {
    if (!h || !x || !out) return fail(h, DS_EINVAL, "ds_tdgsc_process: NULL argument");
    if (h->cfg.algo != DS_ALGO_TDGSC) return fail(h, DS_ESTATE, "ds_tdgsc_process: handle is not a DS_ALGO_TDGSC object");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_tdgsc_process: n_samples must be a multiple of the block length");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, n = n_samples, T = n / h->cfg.hop;
    IoSpec io = {{x, nullptr, nullptr}, {B * M * n * 4, 0, 0}, {out, p, bm, w, nullptr},
                 {B * n * 4, p ? B * T * h->K * 4 : 0, bm ? B * n * (M - 1) * 4 : 0, w ? B * h->cfg.hop * (M - 1) * 4 : 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    rc = tdgsc_run(h, din[0], (long long)(M * n), (long long)n, n_samples, postfilter, dout[0], (long long)n, p ? dout[1] : nullptr,
                   bm ? dout[2] : nullptr, w ? dout[3] : nullptr);
    if (rc) return rc;
    return io_end(h, mem, io, dout);
}

int ds_fdgsc_process(ds_handle* h, const float* x, int n_samples, int postfilter, int dc_notch, float* out, float* p, float* fix_output,
                     float* fix_delayed, float* bm_output, float* aligned, float* aligned_delayed, float* w_aic, float* w_bm, int mem) {
    if (!h || !x || !out) return fail(h, DS_EINVAL, "ds_fdgsc_process: NULL argument");
    if (h->cfg.algo != DS_ALGO_FDGSC) return fail(h, DS_ESTATE, "ds_fdgsc_process: handle is not a DS_ALGO_FDGSC object");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_fdgsc_process: n_samples must be a multiple of the block length");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, n = n_samples, T = n / h->cfg.hop, FL = h->cfg.hop;
    // nine optional results: staged through the handle's scratch slots by hand (IoSpec carries five)
    float* host_out[9] = {out, p, fix_output, fix_delayed, bm_output, aligned, aligned_delayed, w_aic, w_bm};
    const size_t bytes[9] = {B * n * 4, B * T * h->K * 4, B * n * 4, B * n * 4, B * M * n * 4, B * M * n * 4, B * M * n * 4, B * FL * M * 4, B * M * FL * 4};
    if (mem == DS_MEM_DEVICE)
        return fdgsc_run(h, x, (long long)(M * n), (long long)n, n_samples, postfilter, dc_notch, out, (long long)n, p, fix_output, fix_delayed, bm_output,
                         aligned, aligned_delayed, w_aic, w_bm);
    size_t total = B * M * n * 4;
    for (int i = 0; i < 9; ++i) if (host_out[i]) total += bytes[i];
    rc = stage_reserve(h, 0, total); if (rc) return rc;
    char* base = (char*)h->dev_buf[0];
    DS_HIP(h, hipMemcpyAsync(base, x, B * M * n * 4, hipMemcpyHostToDevice, h->stream));
    float* dev_out[9];
    size_t off = B * M * n * 4;
    for (int i = 0; i < 9; ++i) { dev_out[i] = host_out[i] ? (float*)(base + off) : nullptr; if (host_out[i]) off += bytes[i]; }
    rc = fdgsc_run(h, (const float*)base, (long long)(M * n), (long long)n, n_samples, postfilter, dc_notch, dev_out[0], (long long)n, dev_out[1], dev_out[2],
                   dev_out[3], dev_out[4], dev_out[5], dev_out[6], dev_out[7], dev_out[8]);
    if (rc) return rc;
    for (int i = 0; i < 9; ++i)
        if (host_out[i]) DS_HIP(h, hipMemcpyAsync(host_out[i], dev_out[i], bytes[i], hipMemcpyDeviceToHost, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

}  // extern "C"
