// ds_ops.hpp — frame-level operators of the hot path as stand-alone (utterance, bin) programs:
// the L2 objects the reference lets callers drive frame by frame:
//   NoiseEstimationMCRA.estimation   noise_estimation/mcra.py:27-77
//   McMcra.estimation                noise_estimation/mc_mcra.py:179-224
//   NsOmlsaMulti.estimation          noise_estimation/omlsa_multi.py:73-156
//   SubbandLMS / SubbandLmsMc.update adaptivefilter/SubbandLMS.py:28-84, SubbandLmsMc.py:144-191
//   SubbandRLS.update                adaptivefilter/SubbandRLS.py:44-71
// One GPU thread owns one (utterance, bin) and walks the T frames of the call; state is a
// structure-of-arrays [B][NF][KP] of floats so that lane k touches address k: coalesced.
// Written single-source (DS_HD) so tests/emul can run the same code serially on the CPU.
#pragma once
#include "ds_core.hpp"

namespace ds {

enum { OP_MCRA = 0, OP_MCMCRA = 1, OP_OMLSA = 2, OP_SUBLMS = 3, OP_SUBRLS = 4, OP_MCSPPBASE = 5, OP_WPE = 6 };

struct OpParams {
    int B, K, KP, T;          // utterances, bins, padded plane length, frames in this call
    float* st;                // state [B][NF][KP]
    int NF;
    const float* in0;         // op-specific inputs (device)
    const float* in1;
    const float* in2;
    float* out0;              // op-specific outputs (device)
    float* out1;
    float* out2;
    int M;                    // channels: mics (McMcra), beam + references (OMLSA), filter channels (subband LMS)
    int N;                    // filter taps
    int frm_cnt, ell, L;      // MCRA counters before the first frame of the call (uniform over the batch)
    int first_frame;          // OMLSA: 1 until the first estimation() has run
    int in_complex;           // MCRA: input is complex, take |.|^2 (mcra.py:29-30)
    int has_p;                // subband LMS: per-bin update probability given
    int norm;                 // subband LMS: power normalisation on (SubbandAF.py:18)
    float mu, alpha, reg, lam;   // step, power smoothing, regulariser (update(alpha=1e-4)), RLS forgetting factor
};

DS_HD float& st_at(const OpParams& p, int b, int f, int k) { return p.st[((long long)b * p.NF + f) * p.KP + k]; }

// advance the uniform MCRA counters by one frame (mcra.py:52-56,72-74); returns `reset` for this frame
DS_HD bool mcra_tick(int& frm, int& ell, int L) {
    const bool reset = (frm != 0) && (ell % L == 0);
    if (reset) ell = 0;
    return reset;
}

// ------------------------------------------------------------------------------------------------
// MCRA: in0 = Y [B][T][K] power (or complex [B][T][K][2]), out0 = lambda_d [B][T][K].  state: S,Smin,Stmp,p,lambda_d
// ------------------------------------------------------------------------------------------------
DS_HD float mcra_in(const OpParams& p, long long base, int k) {
    if (p.in_complex) { const float re = p.in0[2 * (base + k)], im = p.in0[2 * (base + k) + 1]; return fma_(re, re, im * im); }
    return p.in0[base + k];
}

DS_HD void op_mcra(const OpParams& p, int b, int k) {
    float st[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) st[f] = st_at(p, b, f, k);
    int frm = p.frm_cnt, ell = p.ell;
    for (int t = 0; t < p.T; ++t) {
        const long long base = ((long long)b * p.T + t) * p.K;
        const bool reset = mcra_tick(frm, ell, p.L);
        const float y0 = mcra_in(p, base, k);
        const float ym = k > 0 ? mcra_in(p, base, k - 1) : 0.0f;
        const float yp = k < p.K - 1 ? mcra_in(p, base, k + 1) : 0.0f;
        mcra_bin(st, k, p.K, ym, y0, yp, frm, reset, p.L);
        frm += 1; ell += 1;
        p.out0[base + k] = st[4];
    }
#pragma unroll
    for (int f = 0; f < 5; ++f) st_at(p, b, f, k) = st[f];
}

// ------------------------------------------------------------------------------------------------
// McMcra: in0 = y complex [B][T][K][M]; out0 = p, out1 = G [B][T][K].
// state: Phi_yy, Phi_vv (packed symmetric), then xi, gamma, p, G of the last frame (attributes users read)
// ------------------------------------------------------------------------------------------------
template <int M> DS_HD void op_mcmcra(const OpParams& p, int b, int k) {
    constexpr int NS = M * (M + 1) / 2;
    float pyy[NS], pvv[NS];
#pragma unroll
    for (int f = 0; f < NS; ++f) { pyy[f] = st_at(p, b, f, k); pvv[f] = st_at(p, b, NS + f, k); }
    float pp = 0, G = 0, xi = 0, gam = 0;
    int frm = p.frm_cnt;
    for (int t = 0; t < p.T; ++t) {
        const long long base = (((long long)b * p.T + t) * p.K + k) * M;
        cf Z[M];
#pragma unroll
        for (int m = 0; m < M; ++m) Z[m] = mk(p.in0[2 * (base + m)], p.in0[2 * (base + m) + 1]);
        mcmcra_bin<M>(pyy, pvv, Z, k, frm, pp, G, xi, gam);
        frm += 1;
        const long long ob = ((long long)b * p.T + t) * p.K + k;
        p.out0[ob] = pp;
        p.out1[ob] = G;
    }
#pragma unroll
    for (int f = 0; f < NS; ++f) { st_at(p, b, f, k) = pyy[f]; st_at(p, b, NS + f, k) = pvv[f]; }
    if (p.T > 0) {
        st_at(p, b, 2 * NS + 0, k) = xi; st_at(p, b, 2 * NS + 1, k) = gam;
        st_at(p, b, 2 * NS + 2, k) = pp; st_at(p, b, 2 * NS + 3, k) = G;
    }
}

// ------------------------------------------------------------------------------------------------
// NsOmlsaMulti: in0 = y [B][T][K] (beam power), in1 = u [B][T][K][M-1] (reference powers);
// out0 = lambda_d, out1 = G, out2 = p  [B][T][K].
// state floats: [0,5M) M MCRAs (beam first), then zeta_Y, zeta_U[M-1], lambda_d, gamma, G_H1, G, p, xi_hat, q_hat
// ------------------------------------------------------------------------------------------------
DS_HD int omlsa_nf(int M) { return 5 * M + (M - 1) + 8; }

DS_HD void op_omlsa(const OpParams& p, int b, int k) {
    const int M = p.M, K = p.K, R = M - 1;
    const int o_zy = 5 * M, o_zu = 5 * M + 1, o_s = 5 * M + 1 + R;   // o_s: lambda_d, gamma, G_H1, G, p, xi_hat, q_hat
    int frm = p.frm_cnt, ell = p.ell, first = p.first_frame;
    const float Gmin = 0.06309573444801933f;               // 10^(-12/10)  (omlsa_multi.py:35-36)
    for (int t = 0; t < p.T; ++t) {
        const long long yb = ((long long)b * p.T + t) * K;
        const long long ub = yb * R;
        const bool reset = mcra_tick(frm, ell, p.L);
        const float y0 = p.in0[yb + k];
        const float ym = k > 0 ? p.in0[yb + k - 1] : 0.0f, yp = k < K - 1 ? p.in0[yb + k + 1] : 0.0f;
        // M minima-controlled noise trackers (:83-85)
        float mc[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) mc[f] = st_at(p, b, f, k);
        mcra_bin(mc, k, K, ym, y0, yp, frm, reset, p.L);
#pragma unroll
        for (int f = 0; f < 5; ++f) st_at(p, b, f, k) = mc[f];
        const float MU_Y = mc[4];
        float zu_minus_mu_max = -3.0e38f;
        for (int ch = 0; ch < R; ++ch) {
            const float u0 = p.in1[ub + (long long)k * R + ch];
            const float um = k > 0 ? p.in1[ub + (long long)(k - 1) * R + ch] : 0.0f;
            const float up = k < K - 1 ? p.in1[ub + (long long)(k + 1) * R + ch] : 0.0f;
#pragma unroll
            for (int f = 0; f < 5; ++f) mc[f] = st_at(p, b, 5 * (ch + 1) + f, k);
            mcra_bin(mc, k, K, um, u0, up, frm, reset, p.L);
#pragma unroll
            for (int f = 0; f < 5; ++f) st_at(p, b, 5 * (ch + 1) + f, k) = mc[f];
            float zu;
            if (first) zu = u0;                                                           // :92-93
            else zu = fma_(0.8f, st_at(p, b, o_zu + ch, k), (float)(1.0 - 0.8) * fma_(up, 0.25f, fma_(u0, 0.5f, um * 0.25f)));   // :100
            st_at(p, b, o_zu + ch, k) = zu;
            zu_minus_mu_max = fmaxf_(zu_minus_mu_max, zu - mc[4]);
        }
        frm += 1; ell += 1;
        const long long ob = yb + k;
        if (first) {                                                                       // :87-93
            first = 0;
            st_at(p, b, o_s + 0, k) = y0;
            st_at(p, b, o_zy, k) = y0;
            p.out0[ob] = y0; p.out1[ob] = st_at(p, b, o_s + 3, k); p.out2[ob] = st_at(p, b, o_s + 4, k);
            continue;
        }
        const float zy = fma_(0.8f, st_at(p, b, o_zy, k), (float)(1.0 - 0.8) * fma_(yp, 0.25f, fma_(y0, 0.5f, ym * 0.25f)));   // :98
        st_at(p, b, o_zy, k) = zy;
        float Omega = fmaxf_(zy - MU_Y, 1e-6f) / (fmaxf_(zu_minus_mu_max, 0.01f * MU_Y) + 1e-6f);   // :107-109
        Omega = fminf_(fmaxf_(Omega, 0.1f), 100.0f);
        const float gamma_s = fminf_(y0 / fma_(MU_Y, 1.66f, 1e-6f), 100.0f);                     // :115
        float q;
        if (gamma_s < 1.0f || Omega < 0.3f) q = 1.0f;                                            // :122-129
        else q = fmaxf_((10.0f - gamma_s) / (10.0f - 1.0f), (3.0f - Omega) / (3.0f - 0.3f));
        q = fminf_(fmaxf_(q, 1e-6f), 0.9999998f);                                                // :130
        float lam = st_at(p, b, o_s + 0, k);
        const float gamma_pre = st_at(p, b, o_s + 1, k), gh1_pre = st_at(p, b, o_s + 2, k);
        const float gamma = y0 / fmaxf_(lam, 1e-10f);                                            // :134
        const float xi = fma_(0.921f * gh1_pre * gh1_pre, gamma_pre, (float)(1.0 - 0.921) * fmaxf_(gamma - 1.0f, 0.0f));   // :137
        const float nu = gamma * xi / (1.0f + xi);                                               // :140
        const float gh1 = xi / (1.0f + xi);                                                      // :144
        const float pp = 1.0f / (1.0f + q / (1.0f - q) * (1.0f + xi) * expf(-nu));               // :147
        const float at = fma_((float)(1.0 - 0.85), pp, 0.85f);                                   // Base :57, alpha_d = 0.85
        lam = fma_(at, lam, 1.47f * (1.0f - at) * y0);                                           // :149
        float G = powf(gh1, pp) * powf(Gmin, 1.0f - pp);                                         // :153
        G = fmaxf_(fminf_(G, 1.0f), Gmin);
        st_at(p, b, o_s + 0, k) = lam; st_at(p, b, o_s + 1, k) = gamma; st_at(p, b, o_s + 2, k) = gh1;
        st_at(p, b, o_s + 3, k) = G; st_at(p, b, o_s + 4, k) = pp; st_at(p, b, o_s + 5, k) = xi; st_at(p, b, o_s + 6, k) = q;
        p.out0[ob] = lam; p.out1[ob] = G; p.out2[ob] = pp;
    }
}

// ------------------------------------------------------------------------------------------------
// Subband (N)LMS, single- or multi-channel: in0 = x complex [B][T][K][C], in1 = d complex [B][T][K],
// in2 = p [B][T][K] (optional); out0 = err complex [B][T][K].
// state floats: W [N][C] complex, input_buffer [N][C] complex, P
// ------------------------------------------------------------------------------------------------
DS_HD int sublms_nf(int N, int C) { return 4 * N * C + 1; }

DS_HD void op_sublms(const OpParams& p, int b, int k) {
    const int N = p.N, C = p.M, NC2 = 2 * N * C;
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K + k;
        // shift register (SubbandAF.py:50-51 / SubbandLmsMc.py:62-63)
        for (int n = N - 1; n > 0; --n)
            for (int c = 0; c < 2 * C; ++c) st_at(p, b, NC2 + n * 2 * C + c, k) = st_at(p, b, NC2 + (n - 1) * 2 * C + c, k);
        for (int c = 0; c < C; ++c) {
            st_at(p, b, NC2 + 2 * c, k) = p.in0[2 * (fb * C + c)];
            st_at(p, b, NC2 + 2 * c + 1, k) = p.in0[2 * (fb * C + c) + 1];
        }
        cf out = mk(0.0f, 0.0f);
        float pw = 0.0f;
        for (int i = 0; i < N * C; ++i) {
            const cf w = mk(st_at(p, b, 2 * i, k), st_at(p, b, 2 * i + 1, k));
            const cf x = mk(st_at(p, b, NC2 + 2 * i, k), st_at(p, b, NC2 + 2 * i + 1, k));
            out = cfmac(out, x, w);                        // conj(W) X  (SubbandAF.py:107)
            pw += cabs2(x);
        }
        const float pk = p.has_p ? p.in2[fb] : 1.0f;
        const cf d = mk(p.in1[2 * fb], p.in1[2 * fb + 1]);
        const cf err = mk(fma_(-out.x, pk, d.x), fma_(-out.y, pk, d.y));      // d - out * p  (SubbandLMS.py:66-68)
        float scale = 1.0f;
        if (p.norm) {
            float P = st_at(p, b, 2 * NC2, k);
            P = fma_(p.alpha, P, (1.0f - p.alpha) * (pw / (float)C));           // SubbandLMS.py:72-75 ; /M in SubbandLmsMc.py:174-180
            st_at(p, b, 2 * NC2, k) = P;
            scale = 1.0f / (P + p.reg);
        }
        const float g = 2.0f * p.mu * pk * scale;                              // SubbandAF.py:86
        for (int i = 0; i < N * C; ++i) {
            const cf x = mk(st_at(p, b, NC2 + 2 * i, k), st_at(p, b, NC2 + 2 * i + 1, k));
            const cf gr = cmulc(x, err);                                        // X conj(err)
            st_at(p, b, 2 * i, k) = fma_(g, gr.x, st_at(p, b, 2 * i, k));
            st_at(p, b, 2 * i + 1, k) = fma_(g, gr.y, st_at(p, b, 2 * i + 1, k));
        }
        p.out0[2 * fb] = err.x; p.out0[2 * fb + 1] = err.y;
    }
}

// ------------------------------------------------------------------------------------------------
// Subband RLS: in0 = x complex [B][T][K], in1 = d complex [B][T][K]; out0 = err complex [B][T][K].
// state floats: W [N] complex, input_buffer [N] complex, P [N][N] complex
// ------------------------------------------------------------------------------------------------
constexpr int RLS_NMAX = 4;
DS_HD int subrls_nf(int N) { return 4 * N + 2 * N * N; }

DS_HD void op_subrls(const OpParams& p, int b, int k) {
    const int N = p.N, oX = 2 * N, oP = 4 * N;
    const float lam_inv = 1.0f / p.lam;
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K + k;
        for (int n = N - 1; n > 0; --n) {
            st_at(p, b, oX + 2 * n, k) = st_at(p, b, oX + 2 * (n - 1), k);
            st_at(p, b, oX + 2 * n + 1, k) = st_at(p, b, oX + 2 * (n - 1) + 1, k);
        }
        st_at(p, b, oX, k) = p.in0[2 * fb];
        st_at(p, b, oX + 1, k) = p.in0[2 * fb + 1];
        cf X[RLS_NMAX], num[RLS_NMAX], xhP[RLS_NMAX];
        cf out = mk(0.0f, 0.0f);
        for (int i = 0; i < N; ++i) {
            X[i] = mk(st_at(p, b, oX + 2 * i, k), st_at(p, b, oX + 2 * i + 1, k));
            out = cfmac(out, X[i], mk(st_at(p, b, 2 * i, k), st_at(p, b, 2 * i + 1, k)));
        }
        const cf d = mk(p.in1[2 * fb], p.in1[2 * fb + 1]);
        const cf err = csub(d, out);                                             // SubbandRLS.py:52
        cf den = mk(p.lam, 0.0f);
        for (int i = 0; i < N; ++i) {
            cf a = mk(0.0f, 0.0f), r = mk(0.0f, 0.0f);
            for (int j = 0; j < N; ++j) {
                a = cfma(a, mk(st_at(p, b, oP + 2 * (i * N + j), k), st_at(p, b, oP + 2 * (i * N + j) + 1, k)), X[j]);   // (P x)_i
                r = cfmac(r, mk(st_at(p, b, oP + 2 * (j * N + i), k), st_at(p, b, oP + 2 * (j * N + i) + 1, k)), X[j]);  // (x^H P)_i
            }
            num[i] = a; xhP[i] = r;
            den = cfmac(den, a, X[i]);                                            // + conj(x_i) (P x)_i   :56-60
        }
        for (int i = 0; i < N; ++i) {
            const cf kn = cdiv(num[i], den);
            for (int j = 0; j < N; ++j) {                                        // P = (P - kn x^H P) / lambda   :63
                const int q = oP + 2 * (i * N + j);
                const cf pij = cfnma(mk(st_at(p, b, q, k), st_at(p, b, q + 1, k)), kn, xhP[j]);
                st_at(p, b, q, k) = pij.x * lam_inv;
                st_at(p, b, q + 1, k) = pij.y * lam_inv;
            }
            const cf g = cmulc(kn, err);                                         // conj(err) kn   :65
            st_at(p, b, 2 * i, k) = fma_(2.0f * p.mu, g.x, st_at(p, b, 2 * i, k));
            st_at(p, b, 2 * i + 1, k) = fma_(2.0f * p.mu, g.y, st_at(p, b, 2 * i + 1, k));
        }
        p.out0[2 * fb] = err.x; p.out0[2 * fb + 1] = err.y;
    }
}

// ------------------------------------------------------------------------------------------------
// McSppBase.estimation + compute_pmwf_weight (noise_estimation/mcspp_base.py:262-324,220-240):
// in0 = y complex [B][T][K][M]; out0 = p [B][T][K], out1 = w complex [B][T][K][M] (PMWF weights).
// state floats: Phi_yy (Hermitian packed: M diag + M(M-1)/2 complex), Phi_vv (same), MCRA(5), xi, gamma, p, w[M] complex
// ------------------------------------------------------------------------------------------------
DS_HD int mcsppbase_nf(int M) { return 2 * M * M + 5 + 3 + 2 * M; }

template <int M> DS_HD void op_mcsppbase(const OpParams& p, int b, int k) {
    constexpr int NS = M * (M + 1) / 2, NO = M * (M - 1) / 2;
    float yd[M], yo[2 * NO + 1], vd[M], vo[2 * NO + 1], mc[5];
#pragma unroll
    for (int f = 0; f < M; ++f) { yd[f] = st_at(p, b, f, k); vd[f] = st_at(p, b, M * M + f, k); }
#pragma unroll
    for (int f = 0; f < 2 * NO; ++f) { yo[f] = st_at(p, b, M + f, k); vo[f] = st_at(p, b, M * M + M + f, k); }
#pragma unroll
    for (int f = 0; f < 5; ++f) mc[f] = st_at(p, b, 2 * M * M + f, k);
    int frm = p.frm_cnt, ell = p.ell;
    float xi = 0, gam = 0, pp = 0;
    cf w[M];
#pragma unroll
    for (int m = 0; m < M; ++m) w[m] = mk(0.0f, 0.0f);
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K;
        const long long base = (fb + k) * M;
        cf Z[M];
#pragma unroll
        for (int m = 0; m < M; ++m) Z[m] = mk(p.in0[2 * (base + m)], p.in0[2 * (base + m) + 1]);
        herm_rank1<M>(yd, yo, Z, 0.92f, (float)(1.0 - 0.92));                 // estimate_noisy_psd :86-92
        // real-symmetric inverse of Re(Phi_vv) + 1e-6 I  (:277-279)
        float sv[NS], sx[NS];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) {
                const int q = sym_index(i, j, M);
                if (i == j) { sv[q] = vd[i]; sx[q] = yd[i] - vd[i]; }
                else { const int o = off_index(i, j, M); sv[q] = vo[2 * o]; sx[q] = yo[2 * o] - vo[2 * o]; }
            }
        float Lm[M][M], inv_d[M], Li[M][M], iv[NS];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float s = sym_get<M>(sv, j, j) + 1e-6f;
#pragma unroll
            for (int q = 0; q < j; ++q) s = fma_(-Lm[j][q], Lm[j][q], s);
            s = fmaxf_(s, 1e-30f);
            const float r = 1.0f / sqrtf(s);
            inv_d[j] = r; Lm[j][j] = s * r;
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                float tt = sym_get<M>(sv, i, j);
#pragma unroll
                for (int q = 0; q < j; ++q) tt = fma_(-Lm[i][q], Lm[j][q], tt);
                Lm[i][j] = tt * r;
            }
        }
#pragma unroll
        for (int c = 0; c < M; ++c)
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (i < c) { Li[i][c] = 0.0f; continue; }
                float tt = (i == c) ? 1.0f : 0.0f;
#pragma unroll
                for (int q = c; q < i; ++q) tt = fma_(-Lm[i][q], Li[q][c], tt);
                Li[i][c] = tt * inv_d[i];
            }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) {
                float tt = 0.0f;
#pragma unroll
                for (int q = j; q < M; ++q) tt = fma_(Li[q][i], Li[q][j], tt);
                iv[sym_index(i, j, M)] = tt;
            }
        // xi = trace(inv Re(Phi_xx)) ; gamma = Re(y^H inv Re(Phi_xx) inv y)  (:281-285) ; v = inv y
        float tr = 0.0f;
        cf v[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cf a = mk(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const float e = sym_get<M>(iv, i, j);
                tr = fma_(e, sym_get<M>(sx, i, j), tr);
                a.x = fma_(e, Z[j].x, a.x); a.y = fma_(e, Z[j].y, a.y);
            }
            v[i] = a;
        }
        float g = 0.0f;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j)
                g = fma_(sym_get<M>(sx, i, j), fma_(v[i].x, v[j].x, v[i].y * v[j].y), g);   // Re(conj(v_i) v_j) Pxx_ij
        xi = fminf_(fmaxf_(tr, 1e-6f), 1e6f);                                  // :287
        gam = fminf_(fmaxf_(g, 1e-6f), 1e6f);                                  // :288
        // compute_q (:96-118): MCRA on |y_0|^2, q = sqrt(1 - p_mcra)
        const bool reset = mcra_tick(frm, ell, p.L);
        const float y0 = cabs2(Z[0]);
        float ym = 0.0f, yp = 0.0f;
        if (k > 0) { const long long q = (fb + k - 1) * M; ym = cabs2(mk(p.in0[2 * q], p.in0[2 * q + 1])); }
        if (k < p.K - 1) { const long long q = (fb + k + 1) * M; yp = cabs2(mk(p.in0[2 * q], p.in0[2 * q + 1])); }
        mcra_bin(mc, k, p.K, ym, y0, yp, frm, reset, p.L);
        frm += 1; ell += 1;
        float q = sqrtf(1.0f - mc[3]);
        q = fminf_(fmaxf_(q, 0.01f), 0.99f);
        pp = 1.0f / (1.0f + q / (1.0f - q) * (1.0f + xi) * expf(-1.0f * (gam / (1.0f + xi))));   // :120-135
        pp = fminf_(fmaxf_(pp, 0.01f), 0.99f);
        // PMWF weights from Phi_xx (complex, before the noise update) and the real inverse (:220-240, :293)
        const float wsc = 1.0f / (1.0f + xi);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cf a = mk(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const cf x0 = csub(herm_get<M>(yd, yo, j, 0), herm_get<M>(vd, vo, j, 0));   // Phi_xx[j][0]
                const float e = sym_get<M>(iv, i, j);
                a.x = fma_(e, x0.x, a.x); a.y = fma_(e, x0.y, a.y);
            }
            w[i] = cscale(a, wsc);
        }
        // update_noise_psd (:299-324): alpha_d = 0.92
        const float at = fma_((float)(1.0 - 0.92), pp, 0.92f);
        herm_rank1<M>(vd, vo, Z, at, 1.0f - at);
        const long long ob = fb + k;
        p.out0[ob] = pp;
#pragma unroll
        for (int m = 0; m < M; ++m) { p.out1[2 * (ob * M + m)] = w[m].x; p.out1[2 * (ob * M + m) + 1] = w[m].y; }
    }
#pragma unroll
    for (int f = 0; f < M; ++f) { st_at(p, b, f, k) = yd[f]; st_at(p, b, M * M + f, k) = vd[f]; }
#pragma unroll
    for (int f = 0; f < 2 * NO; ++f) { st_at(p, b, M + f, k) = yo[f]; st_at(p, b, M * M + M + f, k) = vo[f]; }
#pragma unroll
    for (int f = 0; f < 5; ++f) st_at(p, b, 2 * M * M + f, k) = mc[f];
    if (p.T > 0) {
        const int o = 2 * M * M + 5;
        st_at(p, b, o, k) = xi; st_at(p, b, o + 1, k) = gam; st_at(p, b, o + 2, k) = pp;
#pragma unroll
        for (int m = 0; m < M; ++m) { st_at(p, b, o + 3 + 2 * m, k) = w[m].x; st_at(p, b, o + 4 + 2 * m, k) = w[m].y; }
    }
}

// ------------------------------------------------------------------------------------------------
// RLS-based online WPE, frequency-domain core of Wpe.update (dereverberation/awpe.py:129-192) on the STFT grid:
// in0 = x_delayed complex [B][T][K][C] (the frame from `delay` hops ago), in1 = d complex [B][T][K][C] (current frame);
// out0 = err complex [B][T][K][C] (dereverberated frame, all channels).
// state floats: W [C][C*N] complex, input_buffer [C][N] complex, P [C*N][C*N] complex, var
// ------------------------------------------------------------------------------------------------
constexpr int WPE_CNMAX = 16, WPE_CMAX = 8;
DS_HD int wpe_nf(int C, int N) { const int CN = C * N; return 2 * C * CN + 2 * CN + 2 * CN * CN + 1; }

DS_HD void op_wpe(const OpParams& p, int b, int k) {
    const int C = p.M, N = p.N, CN = C * N;
    const int oX = 2 * C * CN, oP = oX + 2 * CN, oV = oP + 2 * CN * CN;
    const float lam = p.lam, lam_inv = 1.0f / p.lam;
    for (int t = 0; t < p.T; ++t) {
        const long long fb = (((long long)b * p.T + t) * p.K + k) * C;
        // buffer_input (:80-102): per channel shift along the taps, newest delayed frame at tap 0
        for (int c = 0; c < C; ++c) {
            for (int n = N - 1; n > 0; --n) {
                st_at(p, b, oX + 2 * (c * N + n), k) = st_at(p, b, oX + 2 * (c * N + n - 1), k);
                st_at(p, b, oX + 2 * (c * N + n) + 1, k) = st_at(p, b, oX + 2 * (c * N + n - 1) + 1, k);
            }
            st_at(p, b, oX + 2 * (c * N), k) = p.in0[2 * (fb + c)];
            st_at(p, b, oX + 2 * (c * N) + 1, k) = p.in0[2 * (fb + c) + 1];
        }
        cf X[WPE_CNMAX], num[WPE_CNMAX], xhP[WPE_CNMAX], err[WPE_CMAX];
        for (int i = 0; i < CN; ++i) X[i] = mk(st_at(p, b, oX + 2 * i, k), st_at(p, b, oX + 2 * i + 1, k));
        float dpow = 0.0f;
        for (int c = 0; c < C; ++c) {                                          // err = d - W^H X  (:158-161)
            cf out = mk(0.0f, 0.0f);
            for (int i = 0; i < CN; ++i)
                out = cfmac(out, X[i], mk(st_at(p, b, 2 * (c * CN + i), k), st_at(p, b, 2 * (c * CN + i) + 1, k)));
            const cf d = mk(p.in1[2 * (fb + c)], p.in1[2 * (fb + c) + 1]);
            err[c] = csub(d, out);
            dpow += cabs2(d);
        }
        float var = st_at(p, b, oV, k);
        var = fma_(0.98f, var, (float)(1.0 - 0.98) * (dpow / (float)C));         // :163-165
        st_at(p, b, oV, k) = var;
        cf den = mk(lam * var, 0.0f);
        for (int i = 0; i < CN; ++i) {
            cf a = mk(0.0f, 0.0f), r = mk(0.0f, 0.0f);
            for (int j = 0; j < CN; ++j) {
                a = cfma(a, mk(st_at(p, b, oP + 2 * (i * CN + j), k), st_at(p, b, oP + 2 * (i * CN + j) + 1, k)), X[j]);   // (P X)_i
                r = cfmac(r, mk(st_at(p, b, oP + 2 * (j * CN + i), k), st_at(p, b, oP + 2 * (j * CN + i) + 1, k)), X[j]);  // (X^H P)_i
            }
            num[i] = a; xhP[i] = r;
            den = cfmac(den, a, X[i]);                                            // :174-180
        }
        for (int i = 0; i < CN; ++i) {
            const cf kn = cdiv(num[i], den);
            for (int j = 0; j < CN; ++j) {                                        // P = (P - kn (X^H P)) / lambda  :183-185
                const int q = oP + 2 * (i * CN + j);
                const cf pij = cfnma(mk(st_at(p, b, q, k), st_at(p, b, q + 1, k)), kn, xhP[j]);
                st_at(p, b, q, k) = pij.x * lam_inv;
                st_at(p, b, q + 1, k) = pij.y * lam_inv;
            }
            for (int c = 0; c < C; ++c) {                                         // W_c += conj(err_c) kn  :188-189
                const cf g = cmulc(kn, err[c]);
                st_at(p, b, 2 * (c * CN + i), k) += g.x;
                st_at(p, b, 2 * (c * CN + i) + 1, k) += g.y;
            }
        }
        for (int c = 0; c < C; ++c) { p.out0[2 * (fb + c)] = err[c].x; p.out0[2 * (fb + c) + 1] = err[c].y; }
    }
}

// dispatch one (b, k) of an operator
DS_HD void run_op(int op, const OpParams& p, int b, int k) {
    if (op == OP_MCRA) op_mcra(p, b, k);
    else if (op == OP_OMLSA) op_omlsa(p, b, k);
    else if (op == OP_SUBLMS) op_sublms(p, b, k);
    else if (op == OP_SUBRLS) op_subrls(p, b, k);
    else if (op == OP_WPE) op_wpe(p, b, k);
    else if (op == OP_MCSPPBASE) {
        switch (p.M) {
            case 2: op_mcsppbase<2>(p, b, k); break;
            case 4: op_mcsppbase<4>(p, b, k); break;
            case 6: op_mcsppbase<6>(p, b, k); break;
            case 8: op_mcsppbase<8>(p, b, k); break;
            default: break;
        }
    } else if (op == OP_MCMCRA) {
        switch (p.M) {
            case 2: op_mcmcra<2>(p, b, k); break;
            case 4: op_mcmcra<4>(p, b, k); break;
            case 6: op_mcmcra<6>(p, b, k); break;
            case 8: op_mcmcra<8>(p, b, k); break;
            default: break;
        }
    }
}

}  // namespace ds
