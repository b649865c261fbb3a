// ds_tdfilter.hpp — sample-wise time-domain adaptive filters (the definitions the subband filters specialise):
//   BaseFilter.update  (NLMS)  adaptivefilter/BaseFilter.py:52-85
//   Rls.update                 adaptivefilter/RLS.py:26-42
// One workgroup per utterance walks the samples of the call in order; taps are spread over the lanes, the two
// inner products of a sample are reduced through LDS in a fixed order (bitwise reproducible, and identical in the
// serial CPU run of tests/emul).  NLMS: up to 1024 taps; RLS: up to 64 taps with P = L x L in LDS, up to 256 taps (the reference's
// example/RLS.ipynb) with P left in device memory (256 KB per utterance: L2-resident while the workgroup walks the samples).
#pragma once
#include "ds_core.hpp"

namespace ds {

enum { TDF_NLMS = 0, TDF_RLS = 1 };
constexpr int TDF_NT = 256, TDF_LMAX = 1024, TDF_RLS_LDS = 64, TDF_RLS_LMAX = 256;

struct TdfParams {
    int B, n, L, mode;
    const float* x;      // [B][n]
    const float* d;      // [B][n]
    float* err;          // [B][n]
    float* w;            // [B][L]
    float* buf;          // [B][L]   input_buffer, tap 0 = newest sample
    float* P;            // [B][L][L] (RLS)
    float mu, eps, p, lam;
    int norm;
};

struct TdfShared {
    float buf[TDF_LMAX];
    float w[TDF_LMAX];
    float part[2][TDF_NT];
    float part2[2][16];
    float P[TDF_RLS_LDS][TDF_RLS_LDS + 1];
    float num[TDF_RLS_LMAX], xtp[TDF_RLS_LMAX], kn[TDF_RLS_LMAX];
    int pos;
};
struct TdfRegs { int unused; };

struct TdfEngine {
    typedef TdfShared Sh;
    typedef TdfRegs Rg;
    static constexpr int NT = TDF_NT;

    // logical tap j lives at physical slot (pos + j) mod L; a new sample moves pos one slot back
    template <class Exec> static DS_HD void run(Exec& ex, const TdfParams& p, int b, Sh& sh) {
        const int L = p.L;
        float* wg = p.w + (long long)b * L;
        float* bg = p.buf + (long long)b * L;
        float* Pg = p.mode == TDF_RLS ? p.P + (long long)b * L * L : nullptr;
        const float* x = p.x + (long long)b * p.n;
        const float* d = p.d + (long long)b * p.n;
        float* e = p.err + (long long)b * p.n;
        const bool p_lds = L <= TDF_RLS_LDS;                         // P in LDS, else in place in device memory
        auto Pat = [&](int r, int c) -> float& { return p_lds ? sh.P[r][c] : Pg[(long long)r * L + c]; };
        ex.phase([&](int tid, Rg&) {
            for (int i = tid; i < L; i += NT) { sh.buf[i] = bg[i]; sh.w[i] = wg[i]; }
            if (p.mode == TDF_RLS && p_lds)
                for (int i = tid; i < L * L; i += NT) sh.P[i / L][i % L] = Pg[i];
            if (tid == 0) sh.pos = 0;
        });
        for (int s = 0; s < p.n; ++s) {
            const int pos = (L - (s % L)) % L;                       // slot of tap 0 BEFORE this sample's insert is pos_prev
            const int np = (pos + L - 1) % L;                        // slot that receives the new sample (becomes tap 0)
            ex.phase([&](int tid, Rg&) { if (tid == 0) sh.buf[np] = x[s]; });
            if (p.mode == TDF_NLMS) {
                ex.phase([&](int tid, Rg&) {                         // partial w.x and x.x over this lane's taps
                    float a = 0.0f, q = 0.0f;
                    for (int j = tid; j < L; j += NT) {
                        const float v = sh.buf[(np + j) % L];
                        a = fma_(sh.w[j], v, a);
                        q = fma_(v, v, q);
                    }
                    sh.part[0][tid] = a; sh.part[1][tid] = q;
                });
                ex.phase([&](int tid, Rg&) {
                    if (tid < 16) {
                        float a = 0.0f, q = 0.0f;
                        for (int i = 0; i < 16; ++i) { a += sh.part[0][tid * 16 + i]; q += sh.part[1][tid * 16 + i]; }
                        sh.part2[0][tid] = a; sh.part2[1][tid] = q;
                    }
                });
                ex.phase([&](int tid, Rg&) {
                    float a = 0.0f, q = 0.0f;
                    for (int i = 0; i < 16; ++i) { a += sh.part2[0][i]; q += sh.part2[1][i]; }
                    const float er = d[s] - a;                                        // BaseFilter.py:73
                    const float g = 2.0f * p.p * p.mu * (p.norm ? er / (q + p.eps) : er);   // :75-82
                    for (int j = tid; j < L; j += NT) sh.w[j] = fma_(g, sh.buf[(np + j) % L], sh.w[j]);
                    if (tid == 0) e[s] = er;
                });
            } else {
                ex.phase([&](int tid, Rg&) {                         // num = P x, xtp = x^T P   (RLS.py:33-38)
                    if (tid < L) {
                        float a = 0.0f, r = 0.0f;
                        for (int j = 0; j < L; ++j) {
                            const float v = sh.buf[(np + j) % L];
                            a = fma_(Pat(tid, j), v, a);
                            r = fma_(v, Pat(j, tid), r);
                        }
                        sh.num[tid] = a; sh.xtp[tid] = r;
                    }
                    if (tid == NT - 1) {                             // prior filter output with the OLD weights (:30)
                        float out = 0.0f;
                        for (int j = 0; j < L; ++j) out = fma_(sh.w[j], sh.buf[(np + j) % L], out);
                        sh.part2[0][0] = out;
                    }
                });
                ex.phase([&](int tid, Rg&) {
                    float den = p.lam;
                    for (int j = 0; j < L; ++j) den = fma_(sh.buf[(np + j) % L], sh.num[j], den);
                    const float er = d[s] - sh.part2[0][0];                           // :30
                    if (tid < L) {
                        const float kn = sh.num[tid] / den;                           // :34
                        sh.kn[tid] = kn;
                        sh.w[tid] = fma_(2.0f * p.mu * er, kn, sh.w[tid]);            // :39-40 (update_coef)
                    }
                    if (tid == 0) e[s] = er;
                });
                ex.phase([&](int tid, Rg&) {                         // P = (P - kn x^T P) / lambda   :37
                    const float li = 1.0f / p.lam;
                    for (int i = tid; i < L * L; i += NT) {
                        const int r = i / L, c = i % L;
                        Pat(r, c) = fma_(-sh.kn[r], sh.xtp[c], Pat(r, c)) * li;
                    }
                });
            }
        }
        ex.phase([&](int tid, Rg&) {
            // store the shift register back in logical order (tap 0 first)
            const int fin = (L - (p.n % L)) % L;
            for (int j = tid; j < L; j += NT) { bg[j] = sh.buf[(fin + j) % L]; wg[j] = sh.w[j]; }
            if (p.mode == TDF_RLS && p_lds)
                for (int i = tid; i < L * L; i += NT) Pg[i] = sh.P[i / L][i % L];
        });
    }
};

}  // namespace ds
