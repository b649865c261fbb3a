// ds_wpe64.hpp — RLS-WPE (the frequency-domain core of Wpe.update, dereverberation/awpe.py:152-189) with the WHOLE recursion in double:
// the accuracy mode of the WPE handles (DS_PARAM_WPE_FP64).
//
// The reference carries P in complex128.  The fp32 kernels (ds_wpe.hpp, ds_wpe_wide.hpp) carry it in fp32, and what bounds their error is
// eps x cond(P): on a stationary, strongly reverberant stream cond(P) reaches 2.6e5 and the dereverberated signal sits at 5e-4 of its own
// RMS from the reference (an experiment that rounds ONLY P to fp32 in an otherwise fp64 recursion gives the same figure; DESIGN.md section
// 4.2a).  This program keeps P, W, the tap buffer and var in double, in HBM between calls and in LDS during one: one workgroup per
// (utterance, bin), the full CN x CN matrix in LDS (103 KB at 80 taps-by-channels: one workgroup per CU), every sum in a fixed order (a call
// of T frames is T one-frame calls bit for bit).  It is several times slower than the fp32 kernels and moves four times their bytes: an
// opt-in for streams where the fp32 recursion's 5e-4 matters, not the bench path.  fp64 multiply-adds issue at the fp32 rate on gfx950
// (scratch/micro/valu_rate.hip), so the arithmetic itself costs what the scalar fp32 form would.
//
// Same recursion as the fp32 kernels, Hermitian-preserving form: g = P x, den = lambda var + Re(x^H g), P <- P / lambda - g g^H / (den lambda),
// W_c += conj(err_c) g / den (awpe.py:172-187; x^H P = (P x)^H for a Hermitian P).  Inputs and outputs are the complex64 spectra of the
// fp32 transforms.
// State per (utterance, bin), doubles: P row-major [CN][CN][2], W [C][CN][2], taps [CN][2] (tap j = c N + n, awpe.py:154), var, 1 pad.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "ds_wpe.hpp"
#include "ds_linalg64.hpp"

namespace ds {

constexpr int WPE64_NT = 128;
DS_HD constexpr long long wpe64_bin_doubles(int C, int N) { return 2LL * (C * N) * (C * N) + 2LL * C * (C * N) + 2LL * (C * N) + 2; }

struct Wpe64Params {
    WpeParams w;            // the fp32 kernels' arguments: shapes, inputs / outputs, the delay ring (state / ustride unused here)
    double* state64;        // utterance b, bin k at state64 + b * ustride64 + k * wpe64_bin_doubles
    long long ustride64;
    double lam64;           // the forgetting factor as the double the caller meant: ds_config carries it as a float, and 0.998f is 0.998 (1 + 2.6e-8)
                            // — forty frames of P / lambda later that is 1e-6 of P against the reference's 0.998.  wpe64_lambda() below
};
// the shortest decimal that rounds to the float (<= 7 significant digits), as a double: 0.998f -> 0.998
inline double wpe64_lambda(float lam) { return decimal_double(lam); }

template <int CNP> struct Wpe64Shared {
    static constexpr int RS = CNP + 1;              // row stride of P in complex doubles: odd, so that rows start in different banks
    cd P[CNP * RS];
    cd W[WPE_CMAX][CNP];
    cd X[CNP], g[CNP];
    cd d[WPE_CMAX], err[WPE_CMAX];
    double var, den;
};
struct Wpe64Regs { cd xnew; };

template <int CNP> struct Wpe64Engine {
    static constexpr int NT = WPE64_NT;
    typedef Wpe64Shared<CNP> Sh;
    typedef Wpe64Regs Rg;
    template <class Exec> static DS_HD void run(Exec& ex, const Wpe64Params& q, int blk, Sh& sh) {
        const WpeParams& p = q.w;
        const int C = p.C, N = p.N, CN = C * N, RS = Sh::RS;
        const long long b = blk / p.K, k = blk - b * p.K;
        double* st = q.state64 + b * q.ustride64 + k * wpe64_bin_doubles(C, N);
        cd* stP = reinterpret_cast<cd*>(st);
        cd* stW = stP + (long long)CN * CN;
        cd* stX = stW + (long long)C * CN;
        double* stV = reinterpret_cast<double*>(stX + CN);
        const double lam = q.lam64, lam_inv = 1.0 / q.lam64;
        const int ring_pos = p.dev_ring_pos ? p.dev_ring_pos[0] : p.ring_pos;
        const long long fstride = (long long)p.K * C;
        const long long io0 = ((b * p.T) * p.K + k) * C;                                   // channel 0 of frame 0 of this bin in [B][T][K][C]
        const long long ring0 = p.ring ? ((b * p.ring_len) * p.K + k) * C : 0;             // ... of ring slot 0 in [B][ring_len][K][C]
        auto delayed = [&](int t, int c) {                                                 // x_delayed[c] of frame t (ds_wpe.hpp's rule, word for word)
            const float* src = p.xd;
            long long f = io0 + (long long)t * fstride;
            if (p.ring != nullptr) {
                const bool in_ring = t < p.ring_len;
                src = in_ring ? p.ring : p.d;
                f = in_ring ? ring0 + (long long)((ring_pos + t) % p.ring_len) * fstride : io0 + (long long)(t - p.ring_len) * fstride;
            }
            return mkd((double)src[2 * (f + c)], (double)src[2 * (f + c) + 1]);
        };
        ex.phase([&](int tid, Rg&) {
            for (int e = tid; e < CN * CN; e += NT) { const int i = e / CN, j = e - i * CN; sh.P[i * RS + j] = stP[e]; }
            for (int e = tid; e < C * CN; e += NT) { const int c = e / CN, j = e - c * CN; sh.W[c][j] = stW[e]; }
            if (tid < CN) sh.X[tid] = stX[tid];
            if (tid == 0) sh.var = stV[0];
        });
        for (int t = 0; t < p.T; ++t) {
            const long long f = io0 + (long long)t * fstride;
            // ---- buffer_input (:80-102): per channel shift along the taps, newest delayed frame at tap 0
            ex.phase([&](int tid, Rg& r) {
                if (tid < CN) { const int c = tid / N; r.xnew = tid == c * N ? delayed(t, c) : sh.X[tid - 1]; }
            });
            ex.phase([&](int tid, Rg& r) {
                if (tid < CN) sh.X[tid] = r.xnew;
                if (tid < C) {
                    const float dre = p.d[2 * (f + tid)], dim = p.d[2 * (f + tid) + 1];
                    sh.d[tid] = mkd((double)dre, (double)dim);
                    if (p.ring != nullptr && t >= p.T - p.ring_len) {                      // one of the last ring_len frames: keep it
                        const long long fr = ring0 + (long long)((ring_pos + t) % p.ring_len) * fstride;
                        p.ring[2 * (fr + tid)] = dre; p.ring[2 * (fr + tid) + 1] = dim;
                    }
                }
            });
            // ---- g = P x (lanes 0 .. CN - 1), err_c = d_c - W_c^H x (lanes CN .. CN + C - 1), var (the last lane)   :156-163,172
            ex.phase([&](int tid, Rg&) {
                if (tid < CN) {
                    cd a = mkd(0.0, 0.0);
                    for (int j = 0; j < CN; ++j) a = cdfma(a, sh.P[tid * RS + j], sh.X[j]);
                    sh.g[tid] = a;
                } else if (tid < CN + C) {
                    const int c = tid - CN;
                    cd o = mkd(0.0, 0.0);
                    for (int j = 0; j < CN; ++j) o = cdfmac(o, sh.X[j], sh.W[c][j]);       // + X_j conj(W_cj)
                    const cd e = cdsub(sh.d[c], o);
                    sh.err[c] = e;
                    p.err[2 * (f + c)] = (float)e.x; p.err[2 * (f + c) + 1] = (float)e.y;
                    if (p.err0 != nullptr && c == 0) { const long long f0 = f / C; p.err0[2 * f0] = (float)e.x; p.err0[2 * f0 + 1] = (float)e.y; }
                } else if (tid == NT - 1) {
                    double dpow = 0.0;
                    for (int c = 0; c < C; ++c) dpow += cdabs2(sh.d[c]);
                    sh.var = fmad_(0.98, sh.var, (1.0 - 0.98) * (dpow / (double)C));
                }
            });
            ex.phase([&](int tid, Rg&) {                                                   // den = lambda var + Re(x^H g)   :173-178
                if (tid == 0) {
                    double den = lam * sh.var;
                    for (int l = 0; l < CN; ++l) den += fmad_(sh.X[l].x, sh.g[l].x, sh.X[l].y * sh.g[l].y);
                    sh.den = den;
                }
            });
            // ---- P <- P / lambda - g g^H / (den lambda)  :181-183;  W_c += conj(err_c) g / den  :186-187
            ex.phase([&](int tid, Rg&) {
                // (digital silence from the first frame on: den = 0 and the reference's gain is 0 / 0; the gain is 0 here, as in the fp32 kernels)
                const double dinv = sh.den == 0.0 ? 0.0 : 1.0 / sh.den, dls = dinv * lam_inv;
                for (int e = tid; e < CN * CN; e += NT) {
                    const int i = e / CN, j = e - i * CN;
                    const cd gi = sh.g[i], gj = sh.g[j], P0 = sh.P[i * RS + j];
                    const cd gg = cdmulc(gi, gj);                                          // g_i conj(g_j)
                    sh.P[i * RS + j] = mkd(fmad_(-gg.x, dls, P0.x * lam_inv), fmad_(-gg.y, dls, P0.y * lam_inv));
                }
                for (int e = tid; e < C * CN; e += NT) {
                    const int c = e / CN, j = e - c * CN;
                    const cd kn = cdscale(sh.g[j], dinv);
                    const cd u = cdmulc(kn, sh.err[c]);                                    // kn conj(err_c)
                    sh.W[c][j] = mkd(sh.W[c][j].x + u.x, sh.W[c][j].y + u.y);
                }
            });
        }
        ex.phase([&](int tid, Rg&) {
            for (int e = tid; e < CN * CN; e += NT) { const int i = e / CN, j = e - i * CN; stP[e] = sh.P[i * RS + j]; }
            for (int e = tid; e < C * CN; e += NT) { const int c = e / CN, j = e - c * CN; stW[e] = sh.W[c][j]; }
            if (tid < CN) stX[tid] = sh.X[tid];
            if (tid == 0) stV[0] = sh.var;
        });
    }
};

}  // namespace ds
