// ds_kernels_ops.hip — stand-alone STFT / ISTFT kernels and the frame-level (utterance, bin) operator
// kernel (MCRA, McMcra, NsOmlsaMulti, subband LMS / RLS) for gfx950.
#include <cstdint>
#include <cstdlib>
#include "ds_kernels.hpp"
#include "ds_ops.hpp"
#include "ds_tdfilter.hpp"
#include "ds_wpe2.hpp"

namespace ds {

template <int NFFT, int M, int OV> __global__ void __launch_bounds__(NFFT / 2) ds_stft_kernel(Params p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    typedef StftEngine<NFFT, M, false, OV> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

template <int NFFT, int M, int OV> __global__ void __launch_bounds__(NFFT / 2) ds_istft_kernel(Params p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { apply_tick(p.tick); apply_tick(p.tick2); apply_tick(p.tick3); }
    typedef IstftEngine<NFFT, M, OV> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

// the analysis of the SubbandGSC chain's front end with McCDR as its per-bin program (StftEngine<.., CDR = true>)
template <int NFFT, int M> __global__ void __launch_bounds__(NFFT / 2) ds_stft_cdr_kernel(Params p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    typedef StftEngine<NFFT, M, true> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
template <int NFFT, int M> hipError_t launch_stft_cdr(const Params& p, int nblocks, hipStream_t stream) {
    hipLaunchKernelGGL((ds_stft_cdr_kernel<NFFT, M>), dim3(nblocks), dim3(NFFT / 2), 0, stream, p);
    return hipGetLastError();
}

#if defined(DS_WITH_SHELVED)
// ... and with the DC notch and the FIR bank in front of it: the chain's whole front end as one kernel (StftEngine<.., FRONT = true>).
// Built, bit-identical to the three kernels it replaces, measured SLOWER (profiles/r04a/cfg5_front_fusion_ab.txt): shelved
template <int NFFT, int M> __global__ void __launch_bounds__(NFFT / 2) ds_front_kernel(Params p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    typedef StftEngine<NFFT, M, true, 2, true> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
template <int NFFT, int M> hipError_t launch_front(const Params& p, int nblocks, hipStream_t stream) {
    hipLaunchKernelGGL((ds_front_kernel<NFFT, M>), dim3(nblocks), dim3(NFFT / 2), 0, stream, p);
    return hipGetLastError();
}
#endif

template <int NFFT, int M, int OV> hipError_t launch_stft(const Params& p, int nblocks, hipStream_t stream) {
    hipLaunchKernelGGL((ds_stft_kernel<NFFT, M, OV>), dim3(nblocks), dim3(NFFT / 2), 0, stream, p);
    return hipGetLastError();
}
template <int NFFT, int M, int OV> hipError_t launch_istft(const Params& p, int nblocks, hipStream_t stream) {
    hipLaunchKernelGGL((ds_istft_kernel<NFFT, M, OV>), dim3(nblocks), dim3(NFFT / 2), 0, stream, p);
    return hipGetLastError();
}

// single-channel transforms: one row per wavefront, four rows per workgroup (StftRowsEngine / IstftRowsEngine)
template <int NFFT> __global__ void __launch_bounds__(256) ds_stft_rows_kernel(Params p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    typedef StftRowsEngine<NFFT> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
template <int NFFT> __global__ void __launch_bounds__(256) ds_istft_rows_kernel(Params p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { apply_tick(p.tick); apply_tick(p.tick2); apply_tick(p.tick3); }
    typedef IstftRowsEngine<NFFT> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
template <int NFFT> hipError_t launch_stft_rows(const Params& p0, int rows, hipStream_t stream) {
    Params p = p0;
    p.rows = rows;
    hipLaunchKernelGGL((ds_stft_rows_kernel<NFFT>), dim3((rows + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError();
}
template <int NFFT> hipError_t launch_istft_rows(const Params& p0, int rows, hipStream_t stream) {
    Params p = p0;
    p.rows = rows;
    hipLaunchKernelGGL((ds_istft_rows_kernel<NFFT>), dim3((rows + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError();
}
KernelInfo lookup_stft_rows(int nfft) {
    KernelInfo ki = {nullptr, 0, 0, 0};
    if (nfft == 512) ki = KernelInfo{&launch_stft_rows<512>, 0, plane_len(512 / 2 + 1), 256};
    if (nfft == 1024) ki = KernelInfo{&launch_stft_rows<1024>, 0, plane_len(1024 / 2 + 1), 256};
    return ki;
}
KernelInfo lookup_istft_rows(int nfft) {
    KernelInfo ki = {nullptr, 0, 0, 0};
    if (nfft == 512) ki = KernelInfo{&launch_istft_rows<512>, 0, plane_len(512 / 2 + 1), 256};
    if (nfft == 1024) ki = KernelInfo{&launch_istft_rows<1024>, 0, plane_len(1024 / 2 + 1), 256};
    return ki;
}

#define DS_FOR_EACH_TSHAPE(X) \
    X(256, 1) X(256, 2) X(256, 3) X(256, 4) X(256, 5) X(256, 6) X(256, 7) X(256, 8) \
    X(512, 1) X(512, 2) X(512, 3) X(512, 4) X(512, 5) X(512, 6) X(512, 7) X(512, 8) \
    X(1024, 1) X(1024, 2) X(1024, 3) X(1024, 4) X(1024, 5) X(1024, 6) X(1024, 7) X(1024, 8)

KernelInfo lookup_stft(int nfft, int M, int ov) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) { KernelInfo ki = {ov == 4 ? &launch_stft<NFFT_, M_, 4> : &launch_stft<NFFT_, M_, 2>, 0, plane_len(NFFT_ / 2 + 1), NFFT_ / 2}; return ki; }
    DS_FOR_EACH_TSHAPE(X)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}
KernelInfo lookup_stft_cdr(int nfft, int M) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) { KernelInfo ki = {&launch_stft_cdr<NFFT_, M_>, 0, plane_len(NFFT_ / 2 + 1), NFFT_ / 2}; return ki; }
    X(256, 4) X(256, 6) X(256, 8) X(512, 4) X(512, 6) X(512, 8) X(1024, 4) X(1024, 6) X(1024, 8)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}
// the fused front end: null launch where the shape has no kernel or cannot hold an L-tap bank's history and windows in its LDS
KernelInfo lookup_front(int nfft, int M, int L) {
#if defined(DS_WITH_SHELVED)
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_ && StftEngine<NFFT_, M_, true, 2, true>::front_fits(L)) { \
        KernelInfo ki = {&launch_front<NFFT_, M_>, 0, plane_len(NFFT_ / 2 + 1), NFFT_ / 2}; return ki; }
    X(512, 4) X(512, 6) X(1024, 4) X(1024, 6)
#undef X
#endif
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}
KernelInfo lookup_istft(int nfft, int M, int ov) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) { KernelInfo ki = {ov == 4 ? &launch_istft<NFFT_, M_, 4> : &launch_istft<NFFT_, M_, 2>, 0, plane_len(NFFT_ / 2 + 1), NFFT_ / 2}; return ki; }
    DS_FOR_EACH_TSHAPE(X)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}

// McSpp's band-averaged prior np.mean(q[fmin:fmax]) (mcspp.py:258-260), q = 1 - Gamma: one value per (utterance, frame), needed by every
// bin.  A workgroup of the flat (utterance, bin) grid touches at most two utterances (KP > 256): one wave per (utterance, frame) fetches
// the band in one go and lane 0 adds it up in bin order — the order of mcspp_qavg(), so the value is the same bit for bit.
constexpr int QAVG_BAND = 64, QAVG_TMAX = 64;
__device__ inline void mcspp_qavg_block(const OpParams& p, int b0, float (*band)[QAVG_BAND], float* qa) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fmin = (int)(500.0 * (2 * (p.K - 1)) / 16000.0), fmax = (int)(2000.0 * (2 * (p.K - 1)) / 16000.0), n = fmax - fmin;
    for (int job = wv; job < 2 * p.T; job += 4) {                          // job = (u, t)
        const int u = job / p.T, t = job - u * p.T, b = b0 + u;
        if (b < p.B) {
            if (lane < n) band[wv][lane] = 1.0f - p.in1[((long long)b * p.T + t) * p.K + fmin + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0) {
                float qsum = 0.0f;
                for (int j = 0; j < n; ++j) qsum += band[wv][j];
                qa[u * p.T + t] = qsum / (float)n;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// (the fused McSpp + blocking-filter operator at 6 microphones is pinned to the two waves per SIMD the plain steady-state build reaches by itself)
// the notebook operator (OP_MCSPP, round 6): pinned to two waves per SIMD at 5 microphones and more — 256 registers and 204 B of scratch against 348
// registers at one wave: + 11 % (profiles/r06a/nb_mvdr_waves_ab.txt; the state in an LDS column instead of registers, for the whole call or for the
// eigen-solve only, leaves the same spills — loop invariants and a few doubles of the estimation core — and is no faster: removed).  At 4
// microphones it reaches two waves by itself (209 registers); pinned to three (148 B scratch) it is 8 % slower, to four 14 %
#ifndef DS_MCSPP_FULL_WAVES
#define DS_MCSPP_FULL_WAVES 2
#endif
#ifndef DS_MCSPP_FULL_WAVES_M4
#define DS_MCSPP_FULL_WAVES_M4 1
#endif
#ifndef DS_MCSPP_WAVES
#define DS_MCSPP_WAVES 1
#endif
constexpr int binop_min_waves(int op, int M) { return (op == OP_MCSPP_STEADY_FAN && M >= 5) ? 2 : (op == OP_MCSPP_STEADY && M >= 5) ? DS_MCSPP_WAVES : (op == OP_MCSPP && M >= 5) ? DS_MCSPP_FULL_WAVES : (op == OP_MCSPP && M == 4) ? DS_MCSPP_FULL_WAVES_M4 : 1; }
template <int OP, int M> __global__ void __launch_bounds__(256, binop_min_waves(OP, M)) ds_binop_kernel(OpParams p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    const long long i0 = (long long)blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
    const int b = (int)(i / p.KP), k = (int)(i - (long long)b * p.KP);
    if constexpr (OP == OP_MCSPP || OP == OP_MCSPP_LEAN || OP == OP_MCSPP_STEADY || OP == OP_MCSPP_STEADY_FAN) {
        const int band_n = (int)(2000.0 * (2 * (p.K - 1)) / 16000.0) - (int)(500.0 * (2 * (p.K - 1)) / 16000.0);
        if (!p.in2 && p.T <= QAVG_TMAX && band_n > 0 && band_n <= QAVG_BAND) {   // otherwise the operator sums the band per bin itself
            __shared__ float band[4][QAVG_BAND];
            __shared__ float qa[2 * QAVG_TMAX];
            const int b0 = (int)(i0 / p.KP);
            mcspp_qavg_block(p, b0, band, qa);
            __syncthreads();
            p.in2 = qa; p.in2_b0 = b0;                                    // the operator reads in2[(b - in2_b0) * T + t]
        }
    }
    if (b >= p.B || k >= p.K) return;
    OpCtx c = make_op_ctx(p, i0);
    if constexpr ((OP == OP_MCSPP_STEADY || OP == OP_MCSPP_STEADY_FAN) && M >= 5) {   // parking space for Phi_vv (op_mcspp_lean): two waves per SIMD at 6 microphones
        __shared__ float park[(M * M + 1 + (OP == OP_MCSPP_STEADY_FAN ? RlsFan<2, M>::NW : 0)) * 256];   // ... and, fused, for the blocking filters between their frames
        c.spill = park + threadIdx.x; c.spill_stride = 256;
    }
    if constexpr (OP == OP_MCSPP_STEADY_FAN) {                              // the blocking filters' planes: a descriptor of their own, from the first
        OpParams q = p;                                                     // instance of the workgroup's first utterance on
        q.st = p.fan_st; q.NF = p.fan_NF; q.B = p.B * M; q.dev_cnt = nullptr;
        const OpCtx c2 = make_op_ctx(q, (i0 / p.KP) * M * p.KP);
        c.fan_ctx = &c2;
        run_op_t<OP, M>(c, b, k);
        return;
    }
    run_op_t<OP, M>(c, b, k);
}

template <int F> __global__ void __launch_bounds__(256) ds_subrls_fan_kernel(OpParams p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    const long long i0 = (long long)blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
    const int u = (int)(i / p.KP), k = (int)(i - (long long)u * p.KP);
    if (u >= p.B / F || k >= p.K) return;
    const OpCtx c = make_op_ctx(p, (i0 / p.KP) * F * p.KP);               // descriptor at the first instance of the workgroup's first utterance
    op_subrls_fan<2, F>(c, u, k);
}

template <int F> __global__ void __launch_bounds__(256) ds_sublms_fan_kernel(OpParams p) {
    if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    const long long i0 = (long long)blockIdx.x * blockDim.x, i = i0 + threadIdx.x;
    const int u = (int)(i / p.KP), k = (int)(i - (long long)u * p.KP);
    if (u >= p.B / F || k >= p.K) return;
    const OpCtx c = make_op_ctx(p, (i0 / p.KP) * F * p.KP);
    op_sublms_fan<2, F>(c, u, k);
}

hipError_t launch_binop(int op, const OpParams& p, hipStream_t stream) {
    if (op == OP_SUBLMS && sublms_fan_ok(p)) {
        const int blocks = (int)(((long long)(p.B / p.x_fan) * p.KP + 255) / 256);
        switch (p.x_fan) {
            case 2: hipLaunchKernelGGL(ds_sublms_fan_kernel<2>, dim3(blocks), dim3(256), 0, stream, p); break;
            case 4: hipLaunchKernelGGL(ds_sublms_fan_kernel<4>, dim3(blocks), dim3(256), 0, stream, p); break;
            case 6: hipLaunchKernelGGL(ds_sublms_fan_kernel<6>, dim3(blocks), dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL(ds_sublms_fan_kernel<8>, dim3(blocks), dim3(256), 0, stream, p); break;
        }
        return hipGetLastError();
    }
    if (op == OP_SUBRLS && subrls_fan_ok(p)) {                             // the instances of an utterance as one thread (see op_subrls_fan)
        const int blocks = (int)(((long long)(p.B / p.x_fan) * p.KP + 255) / 256);
        switch (p.x_fan) {
            case 2: hipLaunchKernelGGL(ds_subrls_fan_kernel<2>, dim3(blocks), dim3(256), 0, stream, p); break;
            case 4: hipLaunchKernelGGL(ds_subrls_fan_kernel<4>, dim3(blocks), dim3(256), 0, stream, p); break;
            case 6: hipLaunchKernelGGL(ds_subrls_fan_kernel<6>, dim3(blocks), dim3(256), 0, stream, p); break;
            default: hipLaunchKernelGGL(ds_subrls_fan_kernel<8>, dim3(blocks), dim3(256), 0, stream, p); break;
        }
        return hipGetLastError();
    }
    const long long total = (long long)p.B * p.KP;
    const int blocks = (int)((total + 255) / 256);
#define X(OP_, M_)                                                                                         \
    if (op == OP_ && (!op_is_matrix(OP_) || p.M == M_)) {                                                    \
        hipLaunchKernelGGL((ds_binop_kernel<OP_, M_>), dim3(blocks), dim3(256), 0, stream, p);               \
        return hipGetLastError();                                                                            \
    }
    DS_FOR_EACH_OP(X)
#undef X
    return hipErrorInvalidValue;
}

// FilterDcNotch16 (ds_ops.hpp td_dcnotch is the definition; same arithmetic, same order).  The recursion is serial in time, so one
// lane owns one (utterance, channel) row and the rows are the only parallelism the arithmetic has: a workgroup takes 32 or 64 rows, and
// its four wavefronts split the work by role.  All 256 lanes move the tiles (32 rows x 256 samples, or 64 x 128: NotchShape) between
// HBM and LDS as 16-byte accesses (1 KB of one row, or 512 B of two, per wave instruction; the next tile's loads in flight behind the
// current tile); wave 0 runs the recursion of the rows in place in LDS, 16 samples per register chunk with the next chunk already loading.
constexpr int NOTCH_NT = 256;
template <int ROWS_, int TS_> struct NotchShape {
    static constexpr int ROWS = ROWS_, TS = TS_, LD = TS + 4;
    static constexpr int LPR = TS / 4;                  // lanes across one row segment (16 bytes each)
    static constexpr int RPW = 64 / LPR;                // rows one wave instruction covers
    static constexpr int RPT = ROWS / (4 * RPW);        // row groups a wave moves per tile: wv, wv + 4, ...
};
template <bool VEC, typename S> __global__ void __launch_bounds__(NOTCH_NT) ds_dcnotch_kernel(TdParams p) {
    __shared__ __attribute__((aligned(16))) float tile[2][S::ROWS][S::LD];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, row0 = blockIdx.x * S::ROWS, rows = p.B * p.M;
    const int lr = lane / S::LPR, col = 4 * (lane % S::LPR);            // this lane's row within a group, and its column
    const double r = p.radius, den2 = notch_den2(p.radius);             // the recursion in double: see td_dcnotch
    const bool rec = wv == 0 && lane < S::ROWS && row0 + lane < rows;         // this lane runs the recursion of row `lane`
    double m0 = 0.0, m1 = 0.0;
    if (rec) { m0 = p.mem[(long long)(row0 + lane) * 2]; m1 = p.mem[(long long)(row0 + lane) * 2 + 1]; }
    // the RPT rows this lane moves: group wv, wv + 4, ... (RPW rows each); column `col`
    const float* src[S::RPT];
    float* dst[S::RPT];
#pragma unroll
    for (int j = 0; j < S::RPT; ++j) {
        const int rl = (wv + 4 * j) * S::RPW + lr;
        const int rr = row0 + rl < rows ? row0 + rl : rows - 1;
        const int b = rr / p.M, m = rr - b * p.M;
        src[j] = (p.x_bstride ? p.x + (long long)b * p.x_bstride + (long long)m * p.x_cstride : p.x + (long long)rr * p.n) + col;
        dst[j] = p.y + (long long)rr * p.n + col;
    }
    vec4 v[S::RPT];
    auto fetch = [&](int s0) {
        const int left = p.n - s0 - col;                                      // samples of the row at or after this lane's column
#pragma unroll
        for (int j = 0; j < S::RPT; ++j) {
            if (VEC) {
                v[j] = left > 0 ? *reinterpret_cast<const vec4*>(src[j] + s0) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            } else {
                v[j].x = left > 0 ? src[j][s0] : 0.0f; v[j].y = left > 1 ? src[j][s0 + 1] : 0.0f;
                v[j].z = left > 2 ? src[j][s0 + 2] : 0.0f; v[j].w = left > 3 ? src[j][s0 + 3] : 0.0f;
            }
        }
    };
    fetch(0);
    int cur = 0;
    for (int s0 = 0; s0 < p.n; s0 += S::TS, cur ^= 1) {
        const int ns = p.n - s0 < S::TS ? p.n - s0 : S::TS;
#pragma unroll
        for (int j = 0; j < S::RPT; ++j) *reinterpret_cast<vec4*>(&tile[cur][(wv + 4 * j) * S::RPW + lr][col]) = v[j];
        __syncthreads();
        if (s0 + S::TS < p.n) fetch(s0 + S::TS);                               // next tile's loads fly behind this tile's recursion
        if (rec) {
            float* row = tile[cur][lane];
            auto step = [&](float vin) { return notch_step(m0, m1, r, den2, vin); };
            const int nfull = ns & ~15;
            vec4 a[4], nx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) nx[q] = a[q] = *reinterpret_cast<const vec4*>(row + 4 * q);
            for (int i = 0; i < nfull; i += 16) {
                if (i + 16 < nfull) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) nx[q] = *reinterpret_cast<const vec4*>(row + i + 16 + 4 * q);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[q].x = step(a[q].x); a[q].y = step(a[q].y); a[q].z = step(a[q].z); a[q].w = step(a[q].w); }
#pragma unroll
                for (int q = 0; q < 4; ++q) { *reinterpret_cast<vec4*>(row + i + 4 * q) = a[q]; a[q] = nx[q]; }
            }
            for (int i = nfull; i < ns; ++i) row[i] = step(row[i]);
        }
        __syncthreads();
        const int left = ns - col;
#pragma unroll
        for (int j = 0; j < S::RPT; ++j) {
            const int rl = (wv + 4 * j) * S::RPW + lr;
            if (row0 + rl >= rows || left <= 0) continue;
            const vec4 o = *reinterpret_cast<const vec4*>(&tile[cur][rl][col]);
            if (VEC) {
                *reinterpret_cast<vec4*>(dst[j] + s0) = o;
            } else {
                dst[j][s0] = o.x;
                if (left > 1) dst[j][s0 + 1] = o.y;
                if (left > 2) dst[j][s0 + 2] = o.z;
                if (left > 3) dst[j][s0 + 3] = o.w;
            }
        }
    }
    if (rec) { p.mem[(long long)(row0 + lane) * 2] = m0; p.mem[(long long)(row0 + lane) * 2 + 1] = m1; }
}
typedef NotchShape<32, 256> NotchS;          // 32 rows x 256 samples per tile
typedef NotchShape<64, 128> NotchW;          // 64 rows x 128 samples: every lane of the recursion wave owns a row

// TimeAlignment FIR bank (td_fir is the definition).  One single-wave block = one utterance x 64 * OPL consecutive outputs, a lane =
// OPL consecutive outputs of every channel (OPL = 8, or 4 for calls of one 256-sample block).  Channels go through the block one at a
// time: the input window of channel m (history from the cache, then x) sits in LDS while the windows of the next two or three channels
// are already on their way from memory into registers, so a block holds two windows (5 KB) instead of M and a CU keeps all its wave slots busy.  The
// window is split into OPL phase rows (sample w at [w % OPL][w / OPL]) so that the lanes' reads are consecutive words for any tap; the
// coefficients sit transposed ([M][L], zero-padded to a multiple of 3 * OPL) and are read 16 bytes at a time.  Taps go in blocks of OPL:
// a block needs the 2 * OPL - 1 samples x[o0 - jb - OPL + 1 .. o0 - jb + OPL - 1], kept as register rows of OPL samples; three rows and
// three coefficient blocks rotate so that the LDS reads of the next block are in flight behind the multiply-adds of this one.
// Every output accumulates its taps in the order j = 0 .. L-1, like
// td_fir.  The block of the last tile also leaves the last L - 1 input samples in the other half of the history ping-pong
// (td_fir_cache).
constexpr int FIR_NT = 64, FIR_MMAX = 16, FIR_LMAX = 120;
template <int OPL> struct FirShape {
    static constexpr int TS = FIR_NT * OPL, PAD = 4 * OPL;                  // zero entries in front: the zero-padded taps past L and the row prefetch read them
    static constexpr int RMIN = (TS + FIR_LMAX - 1 + PAD + OPL - 1) / OPL;  // words per phase row
    static constexpr int UNIT = 64 / OPL;                                   // row length = UNIT x odd: staging stores are conflict-free
    static constexpr int RL = ((RMIN + UNIT - 1) / UNIT | 1) * UNIT;
    static constexpr int LOG = OPL == 8 ? 3 : 2;
    static constexpr int NB = (TS + FIR_LMAX - 1 + FIR_NT - 1) / FIR_NT;    // window entries per lane
    static constexpr int D = OPL == 4 ? 3 : 2;                              // channel windows in flight (registers) ahead of the one in LDS
};
template <int OPL> __global__ void __launch_bounds__(FIR_NT) ds_fir_kernel(TdParams p) {
    typedef FirShape<OPL> S;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int M = p.M, L = p.L, Lp = (L + 3 * OPL - 1) / (3 * OPL) * (3 * OPL), b = blockIdx.y, i0 = blockIdx.x * S::TS, tid = threadIdx.x;
    const int nt = p.n - i0 < S::TS ? p.n - i0 : S::TS;                    // outputs of this tile
    float* cs = lds + 2 * OPL * S::RL;                                     // [M][Lp] (+ OPL slack for the prefetch) behind the two windows [2][OPL][RL]
    auto at = [&](int w) { return (w & (OPL - 1)) * S::RL + (w >> S::LOG); };   // w counts from the first pad entry
    for (int i = tid; i < M * Lp; i += FIR_NT) {
        const int m = i / Lp, j = i - m * Lp;
        cs[i] = j < L ? p.coef[j * M + m] : 0.0f;
    }
    for (int i = tid; i < 2 * S::PAD; i += FIR_NT) lds[(i / S::PAD) * OPL * S::RL + at(i % S::PAD)] = 0.0f;
    if (tid < OPL) cs[M * Lp + tid] = 0.0f;
    const float* xb = p.x + (long long)b * p.n * M;
    const bool swapped = p.dev_parity != nullptr && (p.dev_parity[0] & 1);           // device-resident ping-pong parity (graph replay)
    const float* cache_in = swapped ? p.cache_out : p.cache_in;
    float* cache_out = swapped ? const_cast<float*>(p.cache_in) : p.cache_out;
    const float* cache = cache_in + (long long)b * (L - 1) * M;
    const long long xs_s = p.x_chan_major ? 1 : M, xs_c = p.x_chan_major ? p.n : 1;
    // window entry w = 0 .. nt + L - 2  <->  sample s = i0 + w - (L - 1); s < 0 (first tile only: a tile is longer than the history)
    // comes from the cache
    const int nw = nt + L - 1;
    float vr[S::D][S::NB];
    // branch-free: entries past the window re-read its last entry (and are not stored), history entries select the cache address
    auto fetch = [&](int m, float* v) {
        const float* xm_ = xb + (long long)m * xs_c;
        const float* cm_ = cache + (long long)m * (L - 1);
#pragma unroll
        for (int u = 0; u < S::NB; ++u) {
            const int w0 = u * FIR_NT + tid, w = w0 < nw ? w0 : nw - 1, s_ = i0 + w - (L - 1);
            const float* src = s_ >= 0 ? xm_ + (long long)s_ * xs_s : cm_ + (L - 1 + s_);
            v[u] = *src;
        }
    };
    const int o0 = OPL * tid;                                              // first output of this lane within the tile
    const bool live = o0 < nt, vec = p.y_chan_major && p.n % 4 == 0 && reinterpret_cast<uintptr_t>(p.y) % 16 == 0 && o0 + OPL <= nt;
    const bool keep = cache_out != nullptr && L > 1 && blockIdx.x == gridDim.x - 1;
    float mean[OPL], prev[OPL];
    int ridx[OPL];                                                         // word index of x[i0 + o0 + i] in a window: entry o0 + i + L - 1 + PAD
#pragma unroll
    for (int o = 0; o < OPL; ++o) { mean[o] = 0.0f; prev[o] = 0.0f; ridx[o] = at(L - 1 + S::PAD + o) + tid; }
#pragma unroll
    for (int d = 0; d < S::D; ++d)
        if (d < M) fetch(d, vr[d]);
    auto channel = [&](int m, float* v) {
        float* xs = lds + (m & 1) * OPL * S::RL;
#pragma unroll
        for (int u = 0; u < S::NB; ++u) xs[at(u * FIR_NT + tid + S::PAD)] = v[u];   // entries past the window are never used
        __syncthreads();
        if (m + S::D < M) fetch(m + S::D, v);                              // D channel windows fly behind this channel's taps
        if (keep) {                                                        // history for the next call: the last L - 1 samples
            float* co = cache_out + ((long long)b * M + m) * (L - 1);      // one contiguous row per channel (see TdParams::cache_in)
            for (int i = tid; i < L - 1; i += FIR_NT) co[i] = xs[at(i + nt + S::PAD)];
        }
        if (!live) return;
        const float* cm = cs + m * Lp;
        // row k of the window: H_k[i] = x[i0 + o0 - OPL * k + i] = q[i][4 - k]: the phase of entry i does not depend on k, so a lane keeps
        // one pointer per entry and walks it back by three words per loop iteration (the reads below are pointer + immediate offset)
        const float* q[OPL];
#pragma unroll
        for (int i = 0; i < OPL; ++i) q[i] = xs + ridx[i] - 4;
        auto row = [&](int back, float* h) {                               // back = 4 - (k - k0), k0 = the row q points 4 words below
#pragma unroll
            for (int i = 0; i < OPL; ++i) h[i] = q[i][back];
        };
        float acc[OPL];
#pragma unroll
        for (int o = 0; o < OPL; ++o) acc[o] = 0.0f;
        auto coef = [&](int jb, float* c) {
#pragma unroll
            for (int q = 0; q < OPL / 4; ++q) {
                const vec4 c4 = *reinterpret_cast<const vec4*>(cm + jb + 4 * q);
                c[4 * q] = c4.x; c[4 * q + 1] = c4.y; c[4 * q + 2] = c4.z; c[4 * q + 3] = c4.w;
            }
        };
        auto block = [&](const float* c, const float* hi, const float* lo) {   // OPL taps: x[o - u] = hi[o - u] or lo[OPL + o - u]
#pragma unroll
            for (int u = 0; u < OPL; ++u)                                  // tap order ascending for every output
#pragma unroll
                for (int o = 0; o < OPL; ++o) acc[o] = fma_(c[u], o >= u ? hi[o - u] : lo[OPL + o - u], acc[o]);
        };
        // three rows and three coefficient blocks rotate: block k works on (H_k, H_k+1, c_k) while H_k+2 and c_k+1 are on their way from LDS
        float A[OPL], B[OPL], C[OPL], cA[OPL], cB[OPL], cC[OPL];
        row(4, A); row(3, B); coef(0, cA);
        for (int jb = 0; jb < Lp; jb += 3 * OPL) {
            row(2, C); coef(jb + OPL, cB);
            block(cA, A, B);
            row(1, A); coef(jb + 2 * OPL, cC);
            block(cB, B, C);
            row(0, B); coef(jb + 3 * OPL, cA);
            block(cC, C, A);
#pragma unroll
            for (int i = 0; i < OPL; ++i) q[i] -= 3;
        }
        const int nv = nt - o0;                                            // valid outputs of this lane (>= OPL: all)
        if (vec) {
            float* dst = p.y + ((long long)b * M + m) * p.n + i0 + o0;
#pragma unroll
            for (int q = 0; q < OPL / 4; ++q) *reinterpret_cast<vec4*>(dst + 4 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        } else if (p.y_chan_major) {
            float* dst = p.y + ((long long)b * M + m) * p.n + i0 + o0;
#pragma unroll
            for (int o = 0; o < OPL; ++o) if (o < nv) dst[o] = acc[o];
        } else {
            float* dst = p.y + ((long long)b * p.n + i0 + o0) * M + m;
#pragma unroll
            for (int o = 0; o < OPL; ++o) if (o < nv) dst[(long long)o * M] = acc[o];
        }
        if (p.diff && m > 0) {
            float* dd = p.diff + ((long long)b * p.n + i0 + o0) * (M - 1) + m - 1;
#pragma unroll
            for (int o = 0; o < OPL; ++o) if (o < nv) dd[(long long)o * (M - 1)] = prev[o] - acc[o];
        }
#pragma unroll
        for (int o = 0; o < OPL; ++o) { mean[o] += acc[o]; prev[o] = acc[o]; }
    };
    for (int m0 = 0; m0 < M; m0 += S::D) {
#pragma unroll
        for (int d = 0; d < S::D; ++d)
            if (m0 + d < M) channel(m0 + d, vr[d]);
    }
    if (p.mean && live)
        for (int o = 0; o < OPL && o0 + o < nt; ++o) p.mean[(long long)b * p.n + i0 + o0 + o] = mean[o] / (float)M;
}

hipError_t launch_dcnotch(const TdParams& p, hipStream_t stream) {
    // 16-byte accesses when every row of x and y starts on a 16-byte boundary and holds a multiple of 4 samples
    const bool vec = p.n % 4 == 0 && (reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.y)) % 16 == 0 &&
                     (p.x_bstride ? (p.x_bstride % 4 == 0 && p.x_cstride % 4 == 0) : true);
    // 64 rows x 128 samples per tile where there are rows for it: the recursion wave has a row on every lane, so half as many workgroups hold
    // LDS for as long.  Same arithmetic per row; cfg5 + 2.6 % with 10 s per call, + 1.3 % at one block per call, TDGSC + 0.6 %
    // (profiles/r06a/notch_shape_ab.txt)
    const int rows = p.B * p.M;
    if (vec && rows > NotchS::ROWS) {
        hipLaunchKernelGGL((ds_dcnotch_kernel<true, NotchW>), dim3((rows + NotchW::ROWS - 1) / NotchW::ROWS), dim3(NOTCH_NT), 0, stream, p);
        return hipGetLastError();
    }
    const dim3 grid((rows + NotchS::ROWS - 1) / NotchS::ROWS);
    if (vec) hipLaunchKernelGGL((ds_dcnotch_kernel<true, NotchS>), grid, dim3(NOTCH_NT), 0, stream, p);
    else hipLaunchKernelGGL((ds_dcnotch_kernel<false, NotchS>), grid, dim3(NOTCH_NT), 0, stream, p);
    return hipGetLastError();
}
template <int OPL> static void launch_fir_t(const TdParams& p, hipStream_t stream) {
    typedef FirShape<OPL> S;
    const int Lp = (p.L + 3 * OPL - 1) / (3 * OPL) * (3 * OPL);
    const size_t lds = ((size_t)2 * OPL * S::RL + (size_t)p.M * Lp + OPL) * sizeof(float);
    hipLaunchKernelGGL(ds_fir_kernel<OPL>, dim3((unsigned)((p.n + S::TS - 1) / S::TS), (unsigned)p.B), dim3(FIR_NT), lds, stream, p);
}
hipError_t launch_fir(const TdParams& p, hipStream_t stream) {
    if (p.M > FIR_MMAX || p.L > FIR_LMAX || p.L < 1) return hipErrorInvalidValue;
    // (round 5 tried the channels two at a time as packed pairs — window words {x_m, x_m+1}, taps {c_m, c_m+1}: one v_pk_fma_f32 per two
    // multiply-adds by construction.  Bit-identical, and +1.7 % for cfg5 in both regimes, -0.3 % for the 4-channel chains: the compiler
    // already packs the multiply-adds of neighbouring outputs in this kernel.  scratch/shelved_r05/, profiles/r05a/fir_pairs_ab.txt)
#ifdef DS_ABLATE_CHAIN   // timing experiment only: the 4-outputs-per-lane build (145 registers against 241) on long calls
    { static const bool opl4 = std::getenv("DS_ABL_FIR_OPL4") != nullptr; if (opl4) { launch_fir_t<4>(p, stream); return hipGetLastError(); } }
#endif
    if (p.n > 256) launch_fir_t<8>(p, stream); else launch_fir_t<4>(p, stream);
    return hipGetLastError();
}

__global__ void __launch_bounds__(TDF_NT) ds_tdfilter_kernel(TdfParams p) {
    __shared__ TdfShared sh;
    HipExec<TdfRegs> ex;
    TdfEngine::run(ex, p, (int)blockIdx.x, sh);
}
hipError_t launch_tdfilter(const TdfParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(ds_tdfilter_kernel, dim3(p.B), dim3(TDF_NT), 0, stream, p);
    return hipGetLastError();
}

template <int LPB, int CT = 0, int NTAPS = 0> __global__ void __launch_bounds__(WPE_NT) ds_wpe_kernel(WpeParams p) {
    typedef WpeEngine<LPB, CT, NTAPS> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
// two rows of P per lane (ds_wpe2.hpp) for calls of DS_WPE2_MIN_T frames or more: the shapes with C N = 16 / 8 and a lane per channel.  Hoisted
// lane geometry (HipExec<.., 2>: the bin split, bounds and LDS addresses once per call instead of once per phase and frame: 516 -> 435 vector
// instructions per lane and frame at 8 x 2) inside three waves per SIMD (168 registers).  Bit-identical to the one-row kernels, so the choice
// by call length changes no result; at one frame per call the one-row kernel stays (HBM-bound there: twice the lanes, twice the loads in flight:
// the two-row kernel is 20 % slower at T = 1, 9 % faster at T = 312 on BASELINE config 4, profiles/r05a/cfg4_wpe2_ab.txt)
#ifndef DS_WPE2_HOIST
#define DS_WPE2_HOIST 2
#endif
#ifndef DS_WPE2_MIN_T
#define DS_WPE2_MIN_T 8
#endif
template <int CT, int NTAPS> __global__ void __launch_bounds__(WPE_NT, 3) ds_wpe2_kernel(WpeParams p) {
    typedef WpeEngine2<CT, NTAPS> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg, (CT == 4 && NTAPS == 4) ? 0 : DS_WPE2_HOIST> ex;     // (4 x 4 with hoisted geometry: 60 B of scratch at three waves per SIMD)
    E::run(ex, p, (int)blockIdx.x, sh);
}
hipError_t launch_wpe(const WpeParams& p, int generic, hipStream_t stream) {
    const int lpb = wpe_lanes_per_bin(p.C * p.N), bpw = WPE_NT / lpb;
    const unsigned blocks = (unsigned)(((long long)p.B * p.K + bpw - 1) / bpw);
#ifndef DS_NO_WPE2
    if (!(generic & 1) && p.T >= DS_WPE2_MIN_T) {
#define DS_WPE2_SHAPE(C_, N_) \
        if (p.C == C_ && p.N == N_) { \
            constexpr int bpw2 = WpeEngine2<C_, N_>::BPW; \
            hipLaunchKernelGGL((ds_wpe2_kernel<C_, N_>), dim3((unsigned)(((long long)p.B * p.K + bpw2 - 1) / bpw2)), dim3(WPE_NT), 0, stream, p); \
            return hipGetLastError(); \
        }
        DS_WPE2_SHAPE(8, 2) DS_WPE2_SHAPE(4, 4) DS_WPE2_SHAPE(4, 2)
#undef DS_WPE2_SHAPE
    }
#endif
    // the shapes of the BASELINE config (8 channels x 2 taps), of the reference's notebooks and tests (4 x 2, 2 x 3, 4 x 4, 8 x 1) as
    // compile-time shapes; anything else — and everything with `generic` (DS_WPE_GENERIC=1 at ds_create, the A/B and test switch) — through the generic kernels
    if (!(generic & 1)) {
#define DS_WPE_SHAPE(LPB_, C_, N_) \
        if (p.C == C_ && p.N == N_) { hipLaunchKernelGGL((ds_wpe_kernel<LPB_, C_, N_>), dim3(blocks), dim3(WPE_NT), 0, stream, p); return hipGetLastError(); }
        DS_WPE_SHAPE(16, 8, 2) DS_WPE_SHAPE(8, 4, 2) DS_WPE_SHAPE(16, 4, 4) DS_WPE_SHAPE(8, 8, 1) DS_WPE_SHAPE(8, 2, 3)
#undef DS_WPE_SHAPE
    }
    if (lpb == 4) hipLaunchKernelGGL(ds_wpe_kernel<4>, dim3(blocks), dim3(WPE_NT), 0, stream, p);
    else if (lpb == 8) hipLaunchKernelGGL(ds_wpe_kernel<8>, dim3(blocks), dim3(WPE_NT), 0, stream, p);
    else hipLaunchKernelGGL(ds_wpe_kernel<16>, dim3(blocks), dim3(WPE_NT), 0, stream, p);
    return hipGetLastError();
}

__global__ void __launch_bounds__(64) ds_tick_kernel(TickArgs t) {
    if (threadIdx.x == 0) apply_tick(t);
}
__global__ void __launch_bounds__(64) ds_tick3_kernel(TickArgs a, TickArgs b, TickArgs c) {
    if (threadIdx.x == 0) { apply_tick(a); apply_tick(b); apply_tick(c); }
}
hipError_t launch_tick3(const TickArgs& a, const TickArgs& b, const TickArgs& c, hipStream_t stream) {
    hipLaunchKernelGGL(ds_tick3_kernel, dim3(1), dim3(64), 0, stream, a, b, c);
    return hipGetLastError();
}
hipError_t launch_tick(int* cnt, int frames, int L, int aux_add, int aux_mod, hipStream_t stream) {
    const TickArgs t = {cnt, frames, L > 0 ? L : 1, aux_add, aux_mod};
    hipLaunchKernelGGL(ds_tick_kernel, dim3(1), dim3(64), 0, stream, t);
    return hipGetLastError();
}

// McSpp's band-averaged prior np.mean(q[fmin:fmax]) (mcspp.py:258-260) once per (utterance, frame) instead of once per bin.  One wave
// per row: the lanes fetch the band in one go, lane 0 adds it up in bin order (mcspp_qavg's order, so the value is the same bit for bit).
constexpr int QAVG_MAX = 256;
__global__ void __launch_bounds__(256) ds_mcspp_qavg_kernel(const float* gamma, float* out, int rows, int K, int fmin, int fmax) {
    __shared__ float q[4][QAVG_MAX];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = blockIdx.x * 4 + wv, n = fmax - fmin;
    if (r < rows)
        for (int j = lane; j < n; j += 64) q[wv][j] = 1.0f - gamma[(long long)r * K + fmin + j];
    __syncthreads();
    if (r < rows && lane == 0) {
        float qsum = 0.0f;
        for (int j = 0; j < n; ++j) qsum += q[wv][j];
        out[r] = qsum / (float)n;
    }
}
hipError_t launch_mcspp_qavg(const float* gamma, float* out, int rows, int K, hipStream_t stream) {
    const int fmin = (int)(500.0 * (2 * (K - 1)) / 16000.0), fmax = (int)(2000.0 * (2 * (K - 1)) / 16000.0);
    if (fmax - fmin > QAVG_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ds_mcspp_qavg_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, gamma, out, rows, K, fmin, fmax);
    return hipGetLastError();
}

// realtime wire format (realtime/realtime_processing.py:119-133): int16 LE interleaved [L][C_total] -> float32 / 32768,
// channels [c0, c0 + M) -> x [B][L][M]; enhanced float -> (y * 32768) truncated to int16
__global__ void __launch_bounds__(256) ds_pcm16_to_float_kernel(const short* pcm, float* x, long long n, int Ctot, int c0, int M) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over B * L * M
    if (i >= n) return;
    const long long s = i / M;
    const int m = (int)(i - s * M);
    x[i] = (float)pcm[s * Ctot + c0 + m] / 32768.0f;
}
__global__ void __launch_bounds__(256) ds_float_to_pcm16_kernel(const float* y, short* pcm, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = y[i] * 32768.0f;
    v = fminf(fmaxf(v, -32768.0f), 32767.0f);                                   // astype('<i2') would wrap; saturate instead
    pcm[i] = (short)(int)v;                                                     // truncation toward zero like astype
}
// float32 -> float64 (exact), four samples per lane: the output of ds_process_f64 — the reference's process() returns float64 samples, and widening
// 0.65 GB on one host core (NumPy's astype) was longer than the whole rest of a 10 s call at B = 1024
__global__ void __launch_bounds__(256) ds_float_to_double_kernel(const float* y, double* out, long long n4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    double2* o = reinterpret_cast<double2*>(out) + 2 * i;
    o[0] = make_double2((double)v.x, (double)v.y);
    o[1] = make_double2((double)v.z, (double)v.w);
}
hipError_t launch_float_to_double(const float* y, double* out, long long n, hipStream_t stream) {      // n a multiple of 4 (whole hops)
    hipLaunchKernelGGL(ds_float_to_double_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, stream, y, out, n / 4);
    return hipGetLastError();
}
hipError_t launch_pcm16_to_float(const short* pcm, float* x, long long n, int Ctot, int c0, int M, hipStream_t stream) {
    hipLaunchKernelGGL(ds_pcm16_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pcm, x, n, Ctot, c0, M);
    return hipGetLastError();
}
hipError_t launch_float_to_pcm16(const float* y, short* pcm, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(ds_float_to_pcm16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, y, pcm, n);
    return hipGetLastError();
}

}  // namespace ds
