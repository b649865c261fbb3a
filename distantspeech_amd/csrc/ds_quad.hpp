// ds_quad.hpp — the 8-microphone MVDR bin program spread over a QUAD of lanes (4 lanes per (utterance, bin)).
//
// One thread per bin works for 4 microphones (a 4 x 4 Hermitian solve is ~300 flops and 16 state floats); at 8 microphones the
// same program keeps ~190 values live (64 state floats, the working triangle of the Cholesky sweep, two right-hand sides) and
// spills (ds_frames_kernel<1024, 8, ADAPTIVE>: 256 VGPRs + scratch).  Here a bin belongs to four neighbouring lanes:
// lane l owns rows l and 7 - l of the lower triangle of R (l + (7 - l) = 7 off-diagonal entries + 2 diagonal ones per lane: the
// triangle splits evenly), the right-looking Cholesky sweep of R + diag I runs column by column with the pivot, the column of L
// and the two forward-substitution values of the column handed round by quad broadcasts (DPP quad_perm on the GPU: a full 4-lane
// crossbar in one VALU operand modifier, no LDS), and both substitutions fused into the sweep as in mvdr_output<M>().
// Every cross-lane source is a compile-time lane, every register index is a compile-time index (the rows are padded to their
// longest length, 3 and 7 entries, and updates of the padding are masked), so nothing is indexed dynamically.
//
// The program is written once over a policy Q: on the GPU a value is one lane's float and Q::bcast<S> is a DPP move; the CPU policy
// (tests) carries the four lanes of a quad in one value, so the same text is checked against mvdr_output<8>() without a GPU.
//
// Reference: adaptivebeamformer.py:86-112,119-120 (gated recursive covariance, inverse, MVDR weights, output), beamformer.py:306-336.
#pragma once
#include "ds_core.hpp"

namespace ds {

// ---- policies ---------------------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
struct QuadHip {
    typedef float V;      // one lane's value
    typedef bool B;       // one lane's predicate
    int l;                // lane within the quad
    __device__ explicit QuadHip(int tid) : l(tid & 3) {}
    __device__ B lane_gt(int c) const { return l > c; }
    __device__ B lane_lt(int c) const { return l < c; }
    __device__ B lane_eq(int c) const { return l == c; }
    template <int S> __device__ static V bcast(V x) {                    // value of lane S of the quad, in every lane
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), S | (S << 2) | (S << 4) | (S << 6), 0xf, 0xf, true));
    }
    __device__ static V sel(B m, V a, V b) { return m ? a : b; }
    __device__ static V rsq(V x) { return rsqrtf(x); }
    __device__ static V rcp(V x) { return rcp_(x); }                          // as mvdr_output<M> does
    __device__ static V vmax(V a, float c) { return a > c ? a : c; }
    __device__ static V splat(float c) { return c; }
};
#endif

// the four lanes of a quad side by side (CPU tests of the program text)
struct Q4 {
    float v[4];
};
struct M4 {
    bool v[4];
};
inline Q4 operator+(Q4 a, Q4 b) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
inline Q4 operator-(Q4 a, Q4 b) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
inline Q4 operator*(Q4 a, Q4 b) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] * b.v[i]; return r; }
inline Q4 operator-(Q4 a) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = -a.v[i]; return r; }
inline Q4 fma_(Q4 a, Q4 b, Q4 c) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = __builtin_fmaf(a.v[i], b.v[i], c.v[i]); return r; }
struct QuadCpu {
    typedef Q4 V;
    typedef M4 B;
    B lane_gt(int c) const { M4 m; for (int i = 0; i < 4; ++i) m.v[i] = i > c; return m; }
    B lane_lt(int c) const { M4 m; for (int i = 0; i < 4; ++i) m.v[i] = i < c; return m; }
    B lane_eq(int c) const { M4 m; for (int i = 0; i < 4; ++i) m.v[i] = i == c; return m; }
    template <int S> static V bcast(V x) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = x.v[S]; return r; }
    static V sel(B m, V a, V b) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
    static V rsq(V x) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = 1.0f / sqrtf(x.v[i]); return r; }
    static V rcp(V x) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = 1.0f / x.v[i]; return r; }
    static V vmax(V a, float c) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] > c ? a.v[i] : c; return r; }
    static V splat(float c) { Q4 r; for (int i = 0; i < 4; ++i) r.v[i] = c; return r; }
};

// ---- complex values over V --------------------------------------------------------------------------------------------------------
template <class V> struct cq { V x, y; };
template <class V> DS_HD cq<V> qmk(V a, V b) { cq<V> r; r.x = a; r.y = b; return r; }
template <class V> DS_HD cq<V> qscale(cq<V> a, V s) { return qmk<V>(a.x * s, a.y * s); }
template <class V> DS_HD cq<V> qfnma(cq<V> acc, cq<V> a, cq<V> b) {        // acc - a * b
    return qmk<V>(fma_(-a.x, b.x, fma_(a.y, b.y, acc.x)), fma_(-a.x, b.y, fma_(-a.y, b.x, acc.y)));
}
template <class V> DS_HD cq<V> qfnmac(cq<V> acc, cq<V> a, cq<V> b) {       // acc - a * conj(b)
    return qmk<V>(fma_(-a.x, b.x, fma_(-a.y, b.y, acc.x)), fma_(-a.y, b.x, fma_(a.x, b.y, acc.y)));
}
template <class V> DS_HD cq<V> qfmac(cq<V> acc, cq<V> a, cq<V> b) {        // acc + a * conj(b)
    return qmk<V>(fma_(a.x, b.x, fma_(a.y, b.y, acc.x)), fma_(a.y, b.x, fma_(-a.x, b.y, acc.y)));
}

// ---- the quad's share of one bin's covariance --------------------------------------------------------------------------------------
// lane l: rows i0 = l and i1 = 7 - l of the lower triangle of R: diagonal entries d0, d1 (real) and the entries left of the diagonal,
// r0[k] = R[l][k] (k < l, at most 3) and r1[k] = R[7 - l][k] (k < 7 - l, at most 7); entries at or past the diagonal are padding (zero)
template <class V> struct QuadRows {
    V d0, d1;
    cq<V> r0[3], r1[7];
};
constexpr int quad_owner(int row) { return row < 4 ? row : 7 - row; }   // lane that owns a row
constexpr int quad_slot(int row) { return row < 4 ? 0 : 1; }

// lane l's rows out of / back into the Hermitian-packed state of a bin (StateLayout: 8 real diagonal entries, then the strictly-upper
// entries R[i][j], i < j, row-major, as (re, im) pairs); get(f) / put(f, v) address packed float f of the bin
template <class Get> DS_HD void quad_unpack(int l, Get get, QuadRows<float>& R) {
    R.d0 = get(l); R.d1 = get(7 - l);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool on = k < l;
        const int q = on ? off_index(k, l, 8) : 0;
        R.r0[k].x = on ? get(8 + 2 * q) : 0.0f; R.r0[k].y = on ? -get(8 + 2 * q + 1) : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const bool on = k < 7 - l;
        const int q = on ? off_index(k, 7 - l, 8) : 0;
        R.r1[k].x = on ? get(8 + 2 * q) : 0.0f; R.r1[k].y = on ? -get(8 + 2 * q + 1) : 0.0f;
    }
}
template <class Put> DS_HD void quad_pack(int l, const QuadRows<float>& R, Put put) {
    put(l, R.d0); put(7 - l, R.d1);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < l) { const int q = off_index(k, l, 8); put(8 + 2 * q, R.r0[k].x); put(8 + 2 * q + 1, -R.r0[k].y); }
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (k < 7 - l) { const int q = off_index(k, 7 - l, 8); put(8 + 2 * q, R.r1[k].x); put(8 + 2 * q + 1, -R.r1[k].y); }
}

// own-row picks of a vector every lane holds in full: z[l] and z[7 - l]
template <class Q> DS_HD void quad_pick(const Q& q, const cq<typename Q::V>* z, cq<typename Q::V>& z0, cq<typename Q::V>& z1) {
    z0 = z[0]; z1 = z[7];
#pragma unroll
    for (int c = 1; c < 4; ++c) {
        z0.x = Q::sel(q.lane_eq(c), z[c].x, z0.x); z0.y = Q::sel(q.lane_eq(c), z[c].y, z0.y);
        z1.x = Q::sel(q.lane_eq(c), z[7 - c].x, z1.x); z1.y = Q::sel(q.lane_eq(c), z[7 - c].y, z1.y);
    }
}

// R <- a R + b z z^H on the quad's rows (adaptivebeamformer.py:97-99; herm_rank1<8> on the packed state gives the same numbers:
// every stored element is one fma of the same operands)
template <class Q> DS_HD void quad_rank1(const Q& q, QuadRows<typename Q::V>& R, const cq<typename Q::V>* z, float a, float b) {
    typedef typename Q::V V;
    const V va = Q::splat(a), vb = Q::splat(b);
    cq<V> z0, z1;
    quad_pick(q, z, z0, z1);
    R.d0 = fma_(va, R.d0, vb * fma_(z0.x, z0.x, z0.y * z0.y));
    R.d1 = fma_(va, R.d1, vb * fma_(z1.x, z1.x, z1.y * z1.y));
    // the packed state holds the UPPER element R[k][i] = z_k conj(z_i) (k < i); the row entry R[i][k] is its conjugate, formed from the
    // same products so that both conventions round alike: re = z_k.x z_i.x + z_k.y z_i.y, im(upper) = z_k.y z_i.x - z_k.x z_i.y
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const V re = fma_(z[k].x, z0.x, z[k].y * z0.y), imu = fma_(z[k].y, z0.x, -(z[k].x * z0.y));
        const V nx = fma_(va, R.r0[k].x, vb * re), nyu = fma_(va, -R.r0[k].y, vb * imu);
        R.r0[k].x = Q::sel(q.lane_gt(k), nx, R.r0[k].x);
        R.r0[k].y = Q::sel(q.lane_gt(k), -nyu, R.r0[k].y);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const V re = fma_(z[k].x, z1.x, z[k].y * z1.y), imu = fma_(z[k].y, z1.x, -(z[k].x * z1.y));
        const V nx = fma_(va, R.r1[k].x, vb * re), nyu = fma_(va, -R.r1[k].y, vb * imu);
        R.r1[k].x = Q::sel(q.lane_lt(7 - k), nx, R.r1[k].x);
        R.r1[k].y = Q::sel(q.lane_lt(7 - k), -nyu, R.r1[k].y);
    }
}

// MVDR output  Y = (u^H t) / (u^H u),  u = L^-1 a,  t = L^-1 z,  R + diag I = L L^H  (mvdr_output<8>() spread over the quad).
// Every lane returns the same Y.
template <class Q> DS_HD cq<typename Q::V> quad_mvdr_output(const Q& q, const QuadRows<typename Q::V>& R, float diag,
                                                              const cq<typename Q::V>* a, const cq<typename Q::V>* z) {
    typedef typename Q::V V;
    V d0 = R.d0 + Q::splat(diag), d1 = R.d1 + Q::splat(diag);
    cq<V> r0[3], r1[7];
#pragma unroll
    for (int k = 0; k < 3; ++k) r0[k] = R.r0[k];
#pragma unroll
    for (int k = 0; k < 7; ++k) r1[k] = R.r1[k];
    cq<V> u0, u1, t0, t1;
    quad_pick(q, a, u0, u1);
    quad_pick(q, z, t0, t1);
    V nu = Q::splat(0.0f);
    cq<V> ut = qmk<V>(Q::splat(0.0f), Q::splat(0.0f));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int oj = quad_owner(j);
        const bool s1 = quad_slot(j) == 1;
        // pivot and the column's two forward-substitution values, from the lane that owns row j
        V dj, ujx, ujy, tjx, tjy;
        switch (oj) {
            case 0: dj = Q::template bcast<0>(s1 ? d1 : d0); ujx = Q::template bcast<0>(s1 ? u1.x : u0.x); ujy = Q::template bcast<0>(s1 ? u1.y : u0.y);
                    tjx = Q::template bcast<0>(s1 ? t1.x : t0.x); tjy = Q::template bcast<0>(s1 ? t1.y : t0.y); break;
            case 1: dj = Q::template bcast<1>(s1 ? d1 : d0); ujx = Q::template bcast<1>(s1 ? u1.x : u0.x); ujy = Q::template bcast<1>(s1 ? u1.y : u0.y);
                    tjx = Q::template bcast<1>(s1 ? t1.x : t0.x); tjy = Q::template bcast<1>(s1 ? t1.y : t0.y); break;
            case 2: dj = Q::template bcast<2>(s1 ? d1 : d0); ujx = Q::template bcast<2>(s1 ? u1.x : u0.x); ujy = Q::template bcast<2>(s1 ? u1.y : u0.y);
                    tjx = Q::template bcast<2>(s1 ? t1.x : t0.x); tjy = Q::template bcast<2>(s1 ? t1.y : t0.y); break;
            default: dj = Q::template bcast<3>(s1 ? d1 : d0); ujx = Q::template bcast<3>(s1 ? u1.x : u0.x); ujy = Q::template bcast<3>(s1 ? u1.y : u0.y);
                    tjx = Q::template bcast<3>(s1 ? t1.x : t0.x); tjy = Q::template bcast<3>(s1 ? t1.y : t0.y); break;
        }
        const V r = Q::rsq(Q::vmax(dj, pivot_floor(diag)));      // (mvdr_output()'s floor, word for word)
        const cq<V> uj = qmk<V>(ujx * r, ujy * r), tj = qmk<V>(tjx * r, tjy * r);
        nu = fma_(uj.x, uj.x, fma_(uj.y, uj.y, nu));
        ut = qfmac(ut, tj, uj);                                          // += conj(u_j) t_j
        // column j of L below the diagonal, gathered from the rows' owners (rows j+1 .. 7)
        cq<V> Lc[8];
#pragma unroll
        for (int k = j + 1; k < 8; ++k) {
            const bool ks1 = quad_slot(k) == 1;
            const cq<V> e = ks1 ? r1[j] : r0[j < 3 ? j : 2];             // row k's entry j (slot-0 rows k <= 3 have j <= 2)
            V ex, ey;
            switch (quad_owner(k)) {
                case 0: ex = Q::template bcast<0>(e.x); ey = Q::template bcast<0>(e.y); break;
                case 1: ex = Q::template bcast<1>(e.x); ey = Q::template bcast<1>(e.y); break;
                case 2: ex = Q::template bcast<2>(e.x); ey = Q::template bcast<2>(e.y); break;
                default: ex = Q::template bcast<3>(e.x); ey = Q::template bcast<3>(e.y); break;
            }
            Lc[k] = qmk<V>(ex * r, ey * r);
        }
        // this lane's two rows: slot 0 = row l (active while l > j), slot 1 = row 7 - l (active while 7 - l > j)
        if (j < 3) {
            const typename Q::B on0 = q.lane_gt(j);
            const cq<V> L0 = qmk<V>(r0[j].x * r, r0[j].y * r);            // = Lc[l]
            const cq<V> nu0 = qfnma(u0, L0, uj), nt0 = qfnma(t0, L0, tj);
            u0.x = Q::sel(on0, nu0.x, u0.x); u0.y = Q::sel(on0, nu0.y, u0.y);
            t0.x = Q::sel(on0, nt0.x, t0.x); t0.y = Q::sel(on0, nt0.y, t0.y);
            d0 = Q::sel(on0, fma_(-L0.x, L0.x, fma_(-L0.y, L0.y, d0)), d0);
#pragma unroll
            for (int k = j + 1; k < 3; ++k) {                            // A[l][k] -= L[l][j] conj(L[k][j]),  k < l
                const cq<V> n = qfnmac(r0[k], L0, Lc[k]);
                r0[k].x = Q::sel(q.lane_gt(k), n.x, r0[k].x); r0[k].y = Q::sel(q.lane_gt(k), n.y, r0[k].y);
            }
        }
        if (j < 7) {
            const typename Q::B on1 = q.lane_lt(7 - j);
            const cq<V> L1 = qmk<V>(r1[j].x * r, r1[j].y * r);            // = Lc[7 - l]
            const cq<V> nu1 = qfnma(u1, L1, uj), nt1 = qfnma(t1, L1, tj);
            u1.x = Q::sel(on1, nu1.x, u1.x); u1.y = Q::sel(on1, nu1.y, u1.y);
            t1.x = Q::sel(on1, nt1.x, t1.x); t1.y = Q::sel(on1, nt1.y, t1.y);
            d1 = Q::sel(on1, fma_(-L1.x, L1.x, fma_(-L1.y, L1.y, d1)), d1);
#pragma unroll
            for (int k = j + 1; k < 7; ++k) {                            // A[7-l][k] -= L[7-l][j] conj(L[k][j]),  k < 7 - l
                const cq<V> n = qfnmac(r1[k], L1, Lc[k]);
                r1[k].x = Q::sel(q.lane_lt(7 - k), n.x, r1[k].x); r1[k].y = Q::sel(q.lane_lt(7 - k), n.y, r1[k].y);
            }
        }
    }
    const V inv = Q::rcp(nu);
    return qmk<V>(ut.x * inv, ut.y * inv);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The fused frame kernel for 8 microphones (adaptive MVDR without the Ryy recursion): Engine<NFFT, 8, ALGO_ADAPTIVE, false>'s block
// program with the per-bin phase spread over quads.  One workgroup = one utterance, NT = NFFT / 2 threads.  The transforms, the
// split of the packed spectrum, the MCRA tracker (one thread per bin, its five floats in that thread's registers) and the
// overlap-add are Engine's; the covariance rows of bin k = q * NT / 4 + tid / 4 (q = 0 .. 3) live in the registers of the quad's
// four lanes for the whole call.  The state in HBM is unchanged (Hermitian-packed float4 planes, coalesced 16 B per lane); it is
// dealt to the quads through an LDS staging area in the prologue and collected the same way in the epilogue.  The Nyquist bin
// keeps its packed state in LDS and is run by quad 0 in a fifth pass.  Same numbers as the one-thread program, bit for bit.
// ---------------------------------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
template <int NFFT> struct EngineQ {
    static constexpr int M = 8;
    typedef Engine<NFFT, 8, ALGO_ADAPTIVE, false> EB;
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2, NT = NC, KP = EB::KP, NP = EB::NP;
    static constexpr int BP = NT / 4, PASSES = NC / BP;          // bins per pass, passes over the bins 0 .. NC-1
    static constexpr int SP = 68;                               // floats per bin in the staging area (64 packed + pad, 16-byte rows)
    static constexpr bool WAVE_FFT = EB::WAVE_FFT;
    static constexpr int INV_T0 = EB::INV_T0;
    typedef StateLayout<8, ALGO_ADAPTIVE, false> SL;
    struct Sh : EB::Sh { float pg[K + 3]; };                    // + the MCRA speech presence probability of every bin (the update gate)
    struct Rg {
        typename EB::Rg b;                                      // prefetch registers of Engine's input stream (its Z / st members stay unused)
        float mc[5];                                            // MCRA S, Smin, Stmp, p, lambda_d of bin tid
        QuadRows<float> R[PASSES];
    };
    static_assert(BP * SP * 4 <= (int)sizeof(cf) * 8 * EB::Sh::NCP, "staging area must fit one FFT buffer");

    // the quad's program for one bin
    static __device__ __forceinline__ cf quad_bin(const QuadHip& q, QuadRows<float>& R, const cf* zsrc, int zstride, const cf* steer, int k,
                                                   float pk, const Params& p) {
        cq<float> z[8], a[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) { const cf v = zsrc[m * zstride]; z[m] = qmk<float>(v.x, v.y); }
        const vec4* s4 = reinterpret_cast<const vec4*>(steer + (long long)k * 8);
#pragma unroll
        for (int m = 0; m < 4; ++m) { const vec4 v = s4[m]; a[2 * m] = qmk<float>(v.x, v.y); a[2 * m + 1] = qmk<float>(v.z, v.w); }
        if (pk < p.gate && gate_open(p, k)) quad_rank1(q, R, z, p.alpha_v, p.beta_v);                             // adaptivebeamformer.py:94-99
        cf acc = mk(0.0f, 0.0f);
        if (p.method == METHOD_SRC) {
            acc = cmulc(mk(z[0].x, z[0].y), mk(a[0].x, a[0].y));
        } else if (p.method == METHOD_DS) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc = cfmac(acc, mk(z[m].x, z[m].y), mk(a[m].x, a[m].y));
            acc = cscale(acc, 1.0f / 8);
        } else {
            const cq<float> y = quad_mvdr_output(q, R, p.diag, a, z);
            acc = mk(y.x, y.y);
        }
        return acc;
    }

    template <class Exec> static __device__ __forceinline__ void run(Exec& ex, const Params& p, int blk, Sh& sh) {
        const int b = p.batch0 + blk;
        const long long xb = (long long)blk * p.x_batch_stride;
        const long long yb = (long long)blk * p.y_batch_stride;
        // this utterance's state as Engine lays it out (StateLayout): 17 full planes, then float 68 (lambda_d) as a narrow plane [KP]
        typedef typename EB::SL SLq;
        static_assert(SLq::NPF == 17 && SLq::RT == 1, "8 microphones, no Ryy: 64 covariance floats + 5 MCRA floats");
        float* const ubase = reinterpret_cast<float*>(p.bins) + (long long)b * SLq::ust(KP);
        vec4* bins = reinterpret_cast<vec4*>(ubase);
        float* const btail = ubase + (long long)SLq::NPF * KP * 4;
        float* tin = p.tail_in + (long long)b * M * HOP;
        float* tout = p.tail_out + (long long)b * HOP;
        int* cnt = p.counters + (long long)b * 4;
        const cf* steer = p.steer + (long long)b * p.steer_batch_stride;
        int frm_cnt = cnt[0], ell = cnt[1];
        int old_half = 0;
        cf* fa = &sh.fa[0][0];
        cf* fb = &sh.fb[0][0];
        float* stage = reinterpret_cast<float*>(fa);             // [BP][SP]: free outside the frame loop

        // ---- prologue ------------------------------------------------------------------------------------------------------
        ex.phase([&](int tid, Rg& r) {
            vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
            for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
            const vec4* tin4 = reinterpret_cast<const vec4*>(tin);
            for (int i = tid; i < M * HOP / 4; i += NT) {
                const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                *reinterpret_cast<vec4*>(&sh.xbuf[m][4 * q]) = tin4[i];
            }
            const vec4* tout4 = reinterpret_cast<const vec4*>(tout);
            for (int i = tid; i < HOP / 4; i += NT) *reinterpret_cast<vec4*>(&sh.tail[4 * i]) = tout4[i];
            EB::prefetch_init(p, xb, tid, r.b);
            EB::prefetch(p, xb, 0, tid, r.b);
            // MCRA floats 64 .. 68 of bin tid (planes 16, 17); the Nyquist bin's packed state (all 18 planes) -> LDS
            const vec4 m0 = load_state(&bins[16 * KP + tid]);
            r.mc[0] = m0.x; r.mc[1] = m0.y; r.mc[2] = m0.z; r.mc[3] = m0.w; r.mc[4] = btail[tid];
            if (tid < SLq::NPF) {
                const vec4 v = bins[tid * KP + NC];
                sh.nyq[4 * tid] = v.x; sh.nyq[4 * tid + 1] = v.y; sh.nyq[4 * tid + 2] = v.z; sh.nyq[4 * tid + 3] = v.w;
            } else if (tid == SLq::NPF) {
                sh.nyq[4 * tid] = btail[NC]; sh.nyq[4 * tid + 1] = 0.0f; sh.nyq[4 * tid + 2] = 0.0f; sh.nyq[4 * tid + 3] = 0.0f;
            }
        });
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {                        // covariance planes 0 .. 15 of bins q * BP ..: HBM -> staging -> quads
            ex.phase([&](int tid, Rg&) {
                const int bin = tid % BP, pl0 = tid / BP;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pl = pl0 + 4 * i;
                    *reinterpret_cast<vec4*>(&stage[bin * SP + 4 * pl]) = load_state(&bins[pl * KP + q * BP + bin]);
                }
            });
            ex.phase([&](int tid, Rg& r) {
                const float* src = stage + (tid >> 2) * SP;
                quad_unpack(tid & 3, [&](int f) { return src[f]; }, r.R[q]);
            });
        }

        for (int t = 0; t < p.T; ++t) {
            const int new_half = old_half ^ 1;
            auto ph = [&](bool wave_local, auto f) { if (wave_local) ex.phase_wave(f); else ex.phase(f); };
            ph(WAVE_FFT && p.x_sample_stride == 1, [&](int tid, Rg& r) {
                EB::commit(p, sh, new_half, tid, r.b);
                if (t + 1 < p.T) EB::prefetch(p, xb, t + 1, tid, r.b);
            });
            // ---- forward FFT: 8 packed real transforms (Engine's plan) ----------------------------------------------------
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, true, 0, 1>(tid, NT, sh, nullptr, fa, 1, old_half, M); });
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 1, 2>(tid, NT, sh, fa, fb, 4, 0, M); });
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 2, 0>(tid, NT, sh, fb, fa, 16, 0, M); });
            if (NC == 128) {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, M); });
            } else {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, M); });
                if (NC == 512)
                    ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, -1, false, 0, 0>(tid, NT, sh, fb, fa, 256, 0, M); });
            }
            const cf* F = EB::FWD_FINAL_IS_FB ? fb : fa;
            cf* Zs = EB::FWD_FINAL_IS_FB ? fa : fb;               // [m][NCP]: Z[k][m] of all bins, the Nyquist bin at column NC
            // ---- split: thread k forms Z[k][0..7] -> Zs, |Z_0|^2 -> pw ------------------------------------------------------
            ex.phase([&](int tid, Rg&) {
                const int k = tid, k2 = (NC - k) & (NC - 1);
                const cf w = sh.tb.tw[k];
                float p0 = 0.0f;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const cf A = F[m * Sh::NCP + k], Bc = cconj(F[m * Sh::NCP + k2]);
                    const cf E = cscale(cadd(A, Bc), 0.5f);
                    const cf D = csub(A, Bc);
                    const cf O = mk(0.5f * D.y, -0.5f * D.x);
                    cf Z = cfma(E, w, O);
                    if (k == 0) Z.y = 0.0f;
                    Zs[m * Sh::NCP + k] = Z;
                    if (m == 0) p0 = cabs2(Z);
                    if (k == 0) {                                  // the Nyquist bin of this channel
                        const float zn = A.x - A.y;
                        Zs[m * Sh::NCP + NC] = mk(zn, 0.0f);
                        if (m == 0) sh.pw[NC] = zn * zn;
                    }
                }
                sh.pw[k] = p0;
            });
            // ---- MCRA, one thread per bin (+ the Nyquist bin on its LDS state) -> speech presence probability of every bin ----
            const bool reset = (frm_cnt != 0) && (ell % p.mcra_L == 0);
            ex.phase([&](int tid, Rg& r) {
                const int k = tid;
                mcra_bin(r.mc, k, K, sh.pw[k > 0 ? k - 1 : 0], sh.pw[k], sh.pw[k + 1], frm_cnt, reset, p.mcra_L);
                sh.pg[k] = r.mc[3];
                if (tid == 0) {
                    mcra_bin(sh.nyq + SL::MC_S, NC, K, sh.pw[NC - 1], sh.pw[NC], sh.pw[NC + 1], frm_cnt, reset, p.mcra_L);
                    sh.pg[NC] = sh.nyq[SL::MC_S + 3];
                }
            });
            if (reset) ell = 0;
            frm_cnt += 1; ell += 1;
            // ---- the quads: gated covariance recursion and MVDR output of bins q * BP + tid / 4, then the Nyquist bin ---------
            ex.phase([&](int tid, Rg& r) {
                const QuadHip q(tid);
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps) {
                    const int k = ps * BP + (tid >> 2);
                    cf Yk = quad_bin(q, r.R[ps], Zs + k, Sh::NCP, steer, k, sh.pg[k], p);
                    if (k == 0) Yk.y = 0.0f;                       // irfft ignores Im Y[0] and Im Y[N/2]
                    if ((tid & 3) == 0) sh.Y[k] = Yk;
                }
                if (tid < 4) {
                    QuadRows<float> Rn;
                    quad_unpack(tid, [&](int f) { return sh.nyq[f]; }, Rn);
                    const cf Yn = quad_bin(q, Rn, Zs + NC, Sh::NCP, steer, NC, sh.pg[NC], p);
                    quad_pack(tid, Rn, [&](int f, float v) { sh.nyq[f] = v; });
                    if (tid == 0) sh.Y[NC] = mk(Yn.x, 0.0f);
                }
            });
            // ---- inverse packed real FFT, window, overlap-add (Engine's) ------------------------------------------------------
            ph(WAVE_FFT, [&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, 2, 0, 1>(tid - INV_T0, NT, sh, nullptr, fb, 1, 0, 1); });
            ph(WAVE_FFT, [&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, false, 1, 2>(tid - INV_T0, NT, sh, fb, fa, 4, 0, 1); });
            ph(WAVE_FFT, [&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, false, 2, 0>(tid - INV_T0, NT, sh, fa, fb, 16, 0, 1); });
            if (NC == 128) {
                ex.phase([&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid - INV_T0, NT, sh, fb, fa, 64, 0, 1); });
            } else {
                ex.phase([&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, false, 0, 0>(tid - INV_T0, NT, sh, fb, fa, 64, 0, 1); });
                if (NC == 512)
                    ex.phase([&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid - INV_T0, NT, sh, fa, fb, 256, 0, 1); });
            }
            const cf* Zi = EB::INV_FINAL_IS_FA ? fa : fb;
            ex.phase([&](int tid, Rg&) {
                if (tid < NC / 2) {
                    const int i = tid;
                    const float sc = 1.0f / (float)NC;
                    const float hn = WAVE_FFT ? 0.5f * sh.Y[NC].x : 0.0f;
                    cf z1 = Zi[i], z2 = Zi[i + NC / 2];
                    z1.x += hn; z1.y -= hn; z2.x += hn; z2.y -= hn;
                    const float y0 = sh.tb.win[2 * i] * (z1.x * sc), y1 = sh.tb.win[2 * i + 1] * (z1.y * sc);
                    const float o0 = (y0 + sh.tail[2 * i]) * p.out_scale, o1 = (y1 + sh.tail[2 * i + 1]) * p.out_scale;
                    sh.tail[2 * i] = sh.tb.win[HOP + 2 * i] * (z2.x * sc);
                    sh.tail[2 * i + 1] = sh.tb.win[HOP + 2 * i + 1] * (z2.y * sc);
                    float* dst = p.y + yb + (long long)t * HOP + 2 * i;
                    typedef float f2_t __attribute__((ext_vector_type(2)));
                    f2_t o; o.x = o0; o.y = o1;
                    __builtin_nontemporal_store(o, reinterpret_cast<f2_t*>(dst));
                }
            });
            old_half = new_half;
        }

        // ---- epilogue ------------------------------------------------------------------------------------------------------
        ex.phase([&](int tid, Rg& r) {
            vec4* tin4 = reinterpret_cast<vec4*>(tin);
            for (int i = tid; i < M * HOP / 4; i += NT) {
                const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                store_state(&tin4[i], *reinterpret_cast<const vec4*>(&sh.xbuf[m][old_half * HOP + 4 * q]));
            }
            vec4* tout4 = reinterpret_cast<vec4*>(tout);
            for (int i = tid; i < HOP / 4; i += NT) store_state(&tout4[i], *reinterpret_cast<const vec4*>(&sh.tail[4 * i]));
            vec4 m0;
            m0.x = r.mc[0]; m0.y = r.mc[1]; m0.z = r.mc[2]; m0.w = r.mc[3];
            store_state(&bins[16 * KP + tid], m0); btail[tid] = r.mc[4];
            if (tid < SLq::NPF) {
                vec4 v; v.x = sh.nyq[4 * tid]; v.y = sh.nyq[4 * tid + 1]; v.z = sh.nyq[4 * tid + 2]; v.w = sh.nyq[4 * tid + 3];
                bins[tid * KP + NC] = v;
            } else if (tid == SLq::NPF) {
                btail[NC] = sh.nyq[4 * tid];
            }
            if (tid == 0) { cnt[0] = frm_cnt; cnt[1] = ell; }
        });
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {                        // quads -> staging -> covariance planes 0 .. 15
            ex.phase([&](int tid, Rg& r) {
                float* dst = stage + (tid >> 2) * SP;
                quad_pack(tid & 3, r.R[q], [&](int f, float v) { dst[f] = v; });
            });
            ex.phase([&](int tid, Rg&) {
                const int bin = tid % BP, pl0 = tid / BP;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pl = pl0 + 4 * i;
                    store_state(&bins[pl * KP + q * BP + bin], *reinterpret_cast<const vec4*>(&stage[bin * SP + 4 * pl]));
                }
            });
        }
    }
};
#endif

}  // namespace ds
