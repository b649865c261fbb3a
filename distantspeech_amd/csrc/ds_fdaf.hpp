// ds_fdaf.hpp — overlap-save frequency-domain adaptive filters (block LMS with n_fft = 2 * filter_len):
//   FDAF_PLAIN  FastFreqLms.update                        adaptivefilter/FastFreqLms.py:204-245
//   FDAF_BM     AdaptiveBlockingMatrixFilter.update       beamformer/gsc_bm.py:61-122   (coefficient-clamped)
//   FDAF_AIC    AdaptiveInterferenceCancellation.update   beamformer/gsc_aic.py:53-108  (norm-limited)
// One workgroup per filter instance walks the blocks of the call in order.  All the transforms of a block
// (C + 2 analysis/synthesis FFTs and the 2C..4C FFTs of the gradient / coefficient constraints) run in LDS with the
// packed-real Stockham stages of ds_core.hpp; thread k owns bin k of W, X and P in registers (the Nyquist bin sits
// in LDS and is handled by thread 0), so the only HBM traffic of a block is hop * (C + 2) samples plus, once per
// call, the filter state.  Written against the Exec policy like the frame kernel: tests/emul runs it serially.
#pragma once
#include "ds_core.hpp"

namespace ds {

enum { FDAF_PLAIN = 0, FDAF_BM = 1, FDAF_AIC = 2 };

struct FdafParams {
    int B, T, C;              // filter instances, blocks in this call, input channels
    int kind, constrain, non_causal, weight_norm;
    int trunc;                // fir_truncate, < 0 = None
    int p_mode;               // 0: p = 1, 1: one p per block [B][T], 2: per bin [B][T][K]
    int p_complement;         // use 1 - p (TDGSC.py:154 hands the canceller p = 1 - p)
    int two_path;             // FastFreqLms(two_path=True), plain kind only: a foreground copy of the filter produces the output, the adapting
                              // (background) filter is copied into it whenever the foreground error exceeds the background error by 3 dB
                              // (FastFreqLms.py:94-104,162-176)
    float mu, alpha;
    const float* x;           // [B][T * HOP][C] by default; see the x_* fields
    // input addressing for device-resident chains (all 0 = the dense default): instance b reads the input of instance-group b / x_fan
    // (FDGSC: the M blocking filters of an utterance share the fixed-beamformer output), element (sample s, channel c) of that
    // group at x + group * x_inst_stride + s * x_sample_stride + c * x_chan_stride (channel-major spectra-free hand-offs between stages)
    int x_fan;
    long long x_inst_stride, x_sample_stride, x_chan_stride;
    const float* d;           // [B][T * HOP]
    const float* p;
    float* err;               // [B][T * HOP]
    float* w_out;             // [B][HOP][C] time-domain coefficients after the last block, or null
    float* state;             // per instance: W [C][K] cf | P [K] | previous input block [C][HOP] | tail of d [HOP / 2] | foreground [C][K] cf
    long long state_stride;   // floats between instances (>= fdaf_state_floats)
    const vec4* tables;
};

// floats of state per filter instance (dold: the non_causal delay of filter_len / 2 samples, FastFreqLms.py:85)
DS_HD constexpr long long fdaf_state_floats(int nfft, int C) {
    return 2LL * C * (nfft / 2 + 1) + (nfft / 2 + 1) + (long long)C * (nfft / 2) + nfft / 4 + 2LL * C * (nfft / 2 + 1);
}

template <int NFFT, int CMAX> struct FdafShared {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2;
    static constexpr int NCP = NC + NC / 4;
    cf fa[CMAX][NCP];
    cf fb[CMAX][NCP];
    alignas(16) Tables<NFFT> tb;
    float xold[CMAX][HOP];
    float dold[HOP / 2];
    cf nyqS[CMAX];            // Nyquist bin of the spectra handed to a synthesis transform
    cf nyqW[CMAX];            // Nyquist bin of W, X (real-valued; kept as cf for uniform code), P
    cf nyqX[CMAX];
    cf nyqF[CMAX];            // Nyquist bin of the foreground filter (two_path)
    float nyqP;
    float red[NC];
    float red2[16];
};

template <int CMAX> struct FdafRegs {
    cf W[CMAX];
    cf X[CMAX];
    cf F[CMAX];               // foreground filter (two_path)
    float P;
};

// bin k of the N-point real transform from the NC-point complex transform F of the packed signal (k < NC)
template <int NC> DS_HD cf fdaf_split(const cf* F, int k, const cf* tw) {
    const int k2 = (NC - k) & (NC - 1);
    const cf A = F[k], Bc = cconj(F[k2]);
    const cf E = cscale(cadd(A, Bc), 0.5f);
    const cf D = csub(A, Bc);
    const cf O = mk(0.5f * D.y, -0.5f * D.x);
    cf Z = cfma(E, tw[k], O);
    if (k == 0) Z.y = 0.0f;
    return Z;
}
// packed input of the NC-point inverse transform at index k from bins A = Y[k], B = Y[NC - k]
DS_HD cf fdaf_merge(cf A, cf B, cf w, bool edge) {
    if (edge) { A.y = 0.0f; B.y = 0.0f; }                   // irfft ignores Im Y[0], Im Y[N/2]
    const cf Bc = cconj(B);
    const cf E = cscale(cadd(A, Bc), 0.5f);
    const cf O = cmul(cscale(csub(A, Bc), 0.5f), cconj(w));
    return mk(E.x - O.y, E.y + O.x);
}

template <int NFFT, int CMAX> struct FdafEngine {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2, NT = NC;
    static constexpr bool RES_IS_FB = (NC == 64 || NC == 512);   // where a transform that starts in fa ends
    typedef FdafShared<NFFT, CMAX> Sh;
    typedef FdafRegs<CMAX> Rg;

    // NC-point complex transform of channels [0, C), input unpadded in fa; returns the buffer holding the result
    template <int SIGN, class Exec> static DS_HD cf* fft(Exec& ex, Sh& sh, int C) {
        cf* fa = &sh.fa[0][0];
        cf* fb = &sh.fb[0][0];
        // 256-point plans (filter_len 256): 64 radix-4 butterflies per channel in every stage, i.e. a channel stays inside one wavefront
        // between the stages and those hand-offs need no workgroup barrier (ds_core.hpp Engine::run)
        constexpr bool WAVE_FFT = NC == 256;
        auto ph = [&](bool wave_local, auto f) { if (wave_local) ex.phase_wave(f); else ex.phase(f); };
        ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, CMAX, 4, SIGN, false, 0, 1>(tid, NT, sh, fa, fb, 1, 0, C); });
        ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, CMAX, 4, SIGN, false, 1, 2>(tid, NT, sh, fb, fa, 4, 0, C); });
        ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, CMAX, 4, SIGN, false, 2, 0>(tid, NT, sh, fa, fb, 16, 0, C); });
        if (NC == 128) {
            ex.phase([&](int tid, Rg&) { fft_stage<NFFT, CMAX, 2, SIGN, false, 0, 0>(tid, NT, sh, fb, fa, 64, 0, C); });
        } else if (NC >= 256) {
            ex.phase([&](int tid, Rg&) { fft_stage<NFFT, CMAX, 4, SIGN, false, 0, 0>(tid, NT, sh, fb, fa, 64, 0, C); });
            if (NC == 512)
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, CMAX, 2, SIGN, false, 0, 0>(tid, NT, sh, fa, fb, 256, 0, C); });
        }
        return RES_IS_FB ? fb : fa;
    }

    // spectra src[c][0..NC) (+ sh.nyqS[c]) -> packed synthesis input in fa; pairs (k, NC - k) so that src may be fa
    static DS_HD void merge_pairs(int tid, Sh& sh, const cf* src, int C) {
        cf* fa = &sh.fa[0][0];
        constexpr int NPAIR = NC / 2 + 1;
        for (int idx = tid; idx < C * NPAIR; idx += NT) {
            const int c = idx / NPAIR, k = idx - c * NPAIR;
            const cf* s = src + c * Sh::NCP;
            cf* o = fa + c * Sh::NCP;
            if (k == 0) {
                o[0] = fdaf_merge(s[0], sh.nyqS[c], sh.tb.tw[0], true);
            } else if (k == NC / 2) {
                o[k] = fdaf_merge(s[k], s[k], sh.tb.tw[k], false);
            } else {
                const cf a = s[k], b = s[NC - k];
                o[k] = fdaf_merge(a, b, sh.tb.tw[k], false);
                o[NC - k] = fdaf_merge(b, a, sh.tb.tw[NC - k], false);
            }
        }
    }

    template <class Exec> static DS_HD void run(Exec& ex, const FdafParams& p, int b, Sh& sh) {
        const int C = p.C;
        const long long xs_s = p.x_sample_stride > 0 ? p.x_sample_stride : C, xs_c = p.x_sample_stride > 0 ? p.x_chan_stride : 1;
        const float* xg = p.x + (long long)(b / (p.x_fan > 0 ? p.x_fan : 1)) * (p.x_inst_stride > 0 ? p.x_inst_stride : (long long)p.T * HOP * C);
        const float* dg = p.d + (long long)b * p.T * HOP;
        float* eg = p.err + (long long)b * p.T * HOP;
        float* st = p.state + (long long)b * p.state_stride;
        cf* Wg = reinterpret_cast<cf*>(st);
        float* Pg = st + 2 * C * K;
        float* xog = Pg + K;
        float* dog = xog + C * HOP;
        cf* Fg = reinterpret_cast<cf*>(dog + HOP / 2);
        const bool two_path = p.two_path && p.kind == FDAF_PLAIN && CMAX >= 2;
        cf* fa = &sh.fa[0][0];
        cf* fb = &sh.fb[0][0];
        cf* const res = RES_IS_FB ? fb : fa;       // result of a transform
        cf* const oth = RES_IS_FB ? fa : fb;       // the other buffer: free while `res` is being read
        const float inv_nc = 1.0f / (float)NC;
        const bool td_constrain = p.kind != FDAF_PLAIN && p.constrain;
        const bool rewrite_W = td_constrain || p.trunc >= 0;

        ex.phase([&](int tid, Rg& r) {
            vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
            for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
            for (int i = tid; i < C * HOP; i += NT) sh.xold[i / HOP][i % HOP] = xog[i];
            for (int i = tid; i < HOP / 2; i += NT) sh.dold[i] = dog[i];
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) r.W[c] = Wg[c * K + tid];
            r.P = Pg[tid];
            if (two_path) {
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) r.F[c] = Fg[c * K + tid];
            }
            if (tid == 0) {
                for (int c = 0; c < C; ++c) sh.nyqW[c] = Wg[c * K + NC];
                if (two_path) for (int c = 0; c < C; ++c) sh.nyqF[c] = Fg[c * K + NC];
                sh.nyqP = Pg[NC];
            }
        });

        for (int t = 0; t < p.T; ++t) {
            const float* xt = xg + (long long)t * HOP * xs_s;
            const float* dt = dg + (long long)t * HOP;
            const bool last = t + 1 == p.T;
            const bool want_w = last && p.w_out != nullptr;
            const bool need_td = rewrite_W || want_w;

            // ---- X = rfft([old block | new block])                                        FastFreqLms.py:130-131,155
            ex.phase([&](int tid, Rg&) {
                for (int idx = tid; idx < C * (NC / 2); idx += NT) {
                    const int n = idx / C, c = idx - n * C;
                    const float o0 = sh.xold[c][2 * n], o1 = sh.xold[c][2 * n + 1];
                    const float n0 = xt[(long long)(2 * n) * xs_s + c * xs_c], n1 = xt[(long long)(2 * n + 1) * xs_s + c * xs_c];
                    fa[c * Sh::NCP + n] = mk(o0, o1);
                    fa[c * Sh::NCP + n + NC / 2] = mk(n0, n1);
                    sh.xold[c][2 * n] = n0; sh.xold[c][2 * n + 1] = n1;
                }
            });
            fft<-1>(ex, sh, C);
            // ---- P, and the spectrum of the filter output sum_c X_c W_c                   :156,159
            ex.phase([&](int tid, Rg& r) {
                const int k = tid;
                float pw = 0.0f;
                cf y = mk(0.0f, 0.0f), yf = mk(0.0f, 0.0f);
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) {
                        const cf X = fdaf_split<NC>(res + c * Sh::NCP, k, sh.tb.tw);
                        r.X[c] = X;
                        pw += cabs2(X);
                        y = cadd(y, cmul(X, r.W[c]));
                        if (two_path) yf = cadd(yf, cmul(X, r.F[c]));
                    }
                r.P = fma_(p.alpha, r.P, (1.0f - p.alpha) * pw);
                oth[k] = y;
                if (two_path) oth[Sh::NCP + k] = yf;                          // the foreground filter's output spectrum: a second "channel"
                if (k == 0) {
                    float pn = 0.0f;
                    cf yn = mk(0.0f, 0.0f), yfn = mk(0.0f, 0.0f);
                    for (int c = 0; c < C; ++c) {
                        const cf F0 = res[c * Sh::NCP];
                        const cf X = mk(F0.x - F0.y, 0.0f);
                        sh.nyqX[c] = X;
                        pn += cabs2(X);
                        yn = cadd(yn, cmul(X, sh.nyqW[c]));
                        if (two_path) yfn = cadd(yfn, cmul(X, sh.nyqF[c]));
                    }
                    sh.nyqP = fma_(p.alpha, sh.nyqP, (1.0f - p.alpha) * pn);
                    sh.nyqS[0] = yn;
                    if constexpr (CMAX >= 2) { if (two_path) sh.nyqS[1] = yfn; }      // (launch_fdaf never runs two_path at CMAX = 1)
                }
            });
            const int ny = two_path ? 2 : 1;
            ex.phase([&](int tid, Rg&) { merge_pairs(tid, sh, oth, ny); });
            fft<+1>(ex, sh, ny);
            if (two_path) {
                // ---- transfer logic (FastFreqLms.py:99-104): sum |e_f| against sum |e_b| over the block
                ex.phase([&](int tid, Rg&) {
                    if (tid < NC / 2) {
                        const cf zb = res[tid + NC / 2], zf = res[Sh::NCP + tid + NC / 2];
                        const int s0 = 2 * tid, D = HOP / 2;
                        float d0, d1;
                        if (p.non_causal) {
                            d0 = s0 < D ? sh.dold[s0] : dt[s0 - D];
                            d1 = s0 + 1 < D ? sh.dold[s0 + 1] : dt[s0 + 1 - D];
                        } else {
                            d0 = dt[s0]; d1 = dt[s0 + 1];
                        }
                        sh.red[tid] = fabsf(d0 - zf.x * inv_nc) + fabsf(d1 - zf.y * inv_nc);
                        sh.red[tid + NC / 2] = fabsf(d0 - zb.x * inv_nc) + fabsf(d1 - zb.y * inv_nc);
                    }
                });
                ex.phase([&](int tid, Rg&) {
                    if (tid < 16) {                                               // 8 partial sums each of e_f (0..7) and e_b (8..15), fixed order
                        float a = 0.0f;
                        for (int i = 0; i < NC / 16; ++i) a += sh.red[tid * (NC / 16) + i];
                        sh.red2[tid] = a;
                    }
                });
            }
            // ---- e = d (delayed when non_causal) - last hop of y; E = rfft([0 | e])       :160-172,183-185
            ex.phase([&](int tid, Rg& r) {
                bool transfer = false;
                if (two_path) {
                    float ef = 0.0f, eb = 0.0f;
                    for (int i = 0; i < 8; ++i) { ef += sh.red2[i]; eb += sh.red2[8 + i]; }
                    transfer = 10.0f * log10f(ef / (eb + 1e-6f) + 1e-6f) > 3.0f;    // :101
                    if (transfer) {                                               // foreground <- background (:102)
#pragma unroll
                        for (int c = 0; c < CMAX; ++c)
                            if (c < C) r.F[c] = r.W[c];
                        if (tid == 0) for (int c = 0; c < C; ++c) sh.nyqF[c] = sh.nyqW[c];
                    }
                }
                if (tid < NC / 2) {
                    cf z = res[tid + NC / 2];
                    if (two_path) {                                               // the output is the foreground's; cross-faded on a transfer (:104)
                        const cf zf = res[Sh::NCP + tid + NC / 2];
                        if (transfer) {
                            const float wa0 = sh.tb.win[HOP + 2 * tid], wa1 = sh.tb.win[HOP + 2 * tid + 1];      // sqrt-Hann: window = win^2
                            const float wb0 = sh.tb.win[2 * tid], wb1 = sh.tb.win[2 * tid + 1];
                            z = mk(fma_(wa0 * wa0, zf.x, (wb0 * wb0) * z.x), fma_(wa1 * wa1, zf.y, (wb1 * wb1) * z.y));
                        } else {
                            z = zf;
                        }
                    }
                    const int s0 = 2 * tid, D = HOP / 2;
                    float d0, d1;
                    if (p.non_causal) {
                        d0 = s0 < D ? sh.dold[s0] : dt[s0 - D];
                        d1 = s0 + 1 < D ? sh.dold[s0 + 1] : dt[s0 + 1 - D];
                    } else {
                        d0 = dt[s0]; d1 = dt[s0 + 1];
                    }
                    const float e0 = d0 - z.x * inv_nc, e1 = d1 - z.y * inv_nc;
                    eg[(long long)t * HOP + s0] = e0;
                    eg[(long long)t * HOP + s0 + 1] = e1;
                    fa[tid + NC / 2] = mk(e0, e1);
                } else {
                    fa[tid - NC / 2] = mk(0.0f, 0.0f);
                }
            });
            fft<-1>(ex, sh, 1);
            // ---- gradient conj(X) E / P                                                   :187-188
            const bool grad_constrain = p.kind == FDAF_PLAIN && p.constrain;
            const float two = p.kind == FDAF_PLAIN ? 2.0f : 1.0f;                // :235 vs gsc_bm.py:91 / gsc_aic.py:79
            auto coef = [&](int k) {
                float pk = 1.0f;
                if (p.p_mode == 1) pk = p.p[(long long)b * p.T + t];
                else if (p.p_mode == 2) pk = p.p[((long long)b * p.T + t) * K + k];
                if (p.p_complement) pk = 1.0f - pk;
                return pk * two * p.mu;
            };
            ex.phase([&](int tid, Rg& r) {
                const int k = tid;
                if (p.non_causal && tid < HOP / 2) sh.dold[tid] = dt[HOP / 2 + tid];
                const cf E = fdaf_split<NC>(res, k, sh.tb.tw);
                r.P = fmaxf_(r.P, 1e-4f);
                const float ck = coef(k);
                float nrm = 0.0f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) {
                        const cf g0 = cmul(cconj(r.X[c]), E);
                        const cf g = mk(g0.x / r.P, g0.y / r.P);
                        if (grad_constrain) {
                            oth[c * Sh::NCP + k] = g;
                        } else {
                            r.W[c] = mk(fma_(ck, g.x, r.W[c].x), fma_(ck, g.y, r.W[c].y));
                            if (need_td) oth[c * Sh::NCP + k] = r.W[c];
                            nrm += cabs2(r.W[c]);
                        }
                    }
                if (k == 0) {
                    const cf F0 = res[0];
                    const cf En = mk(F0.x - F0.y, 0.0f);
                    sh.nyqP = fmaxf_(sh.nyqP, 1e-4f);
                    const float cn = coef(NC);
                    for (int c = 0; c < C; ++c) {
                        const cf g0 = cmul(cconj(sh.nyqX[c]), En);
                        const cf g = mk(g0.x / sh.nyqP, g0.y / sh.nyqP);
                        if (grad_constrain) {
                            sh.nyqS[c] = g;
                        } else {
                            sh.nyqW[c] = mk(fma_(cn, g.x, sh.nyqW[c].x), fma_(cn, g.y, sh.nyqW[c].y));
                            sh.nyqS[c] = sh.nyqW[c];
                            nrm += cabs2(sh.nyqW[c]);
                        }
                    }
                }
                sh.red[tid] = nrm;
            });
            if (grad_constrain) {
                // ---- gradient constraint: irfft, zero the last hop, rfft                  :194-198, then W += p 2 mu grad :235
                ex.phase([&](int tid, Rg&) { merge_pairs(tid, sh, oth, C); });
                fft<+1>(ex, sh, C);
                ex.phase([&](int tid, Rg&) {
                    for (int idx = tid; idx < C * NC; idx += NT) {
                        const int c = idx / NC, i = idx - c * NC;
                        fa[c * Sh::NCP + i] = i < NC / 2 ? cscale(res[c * Sh::NCP + i], inv_nc) : mk(0.0f, 0.0f);
                    }
                });
                fft<-1>(ex, sh, C);
                ex.phase([&](int tid, Rg& r) {
                    const int k = tid;
                    const float ck = coef(k);
#pragma unroll
                    for (int c = 0; c < CMAX; ++c)
                        if (c < C) {
                            const cf g = fdaf_split<NC>(res + c * Sh::NCP, k, sh.tb.tw);
                            r.W[c] = mk(fma_(ck, g.x, r.W[c].x), fma_(ck, g.y, r.W[c].y));
                            if (need_td) oth[c * Sh::NCP + k] = r.W[c];
                        }
                    if (k == 0) {
                        const float cn = coef(NC);
                        for (int c = 0; c < C; ++c) {
                            const cf F0 = res[c * Sh::NCP];
                            const cf g = mk(F0.x - F0.y, 0.0f);
                            sh.nyqW[c] = mk(fma_(cn, g.x, sh.nyqW[c].x), fma_(cn, g.y, sh.nyqW[c].y));
                            sh.nyqS[c] = sh.nyqW[c];
                        }
                    }
                });
            }
            if (!need_td) continue;
            // ---- coefficient-domain pass: w = irfft(W) [* norm], constraint / clamp / truncation, W = rfft(w)
            const bool use_norm = p.kind == FDAF_AIC && p.weight_norm;
            if (use_norm)
                ex.phase([&](int tid, Rg&) {
                    if (tid < 16) {
                        float a = 0.0f;
                        for (int i = 0; i < NC / 16; ++i) a += sh.red[tid * (NC / 16) + i];
                        sh.red2[tid] = a;
                    }
                });
            ex.phase([&](int tid, Rg&) { merge_pairs(tid, sh, oth, C); });
            fft<+1>(ex, sh, C);
            ex.phase([&](int tid, Rg&) {
                float norm = 1.0f;
                if (use_norm) {                                                   // gsc_aic.py:81-88
                    float a = 0.0f;
                    for (int i = 0; i < 16; ++i) a += sh.red2[i];
                    const float nv = a / (float)N / (float)N;
                    if (nv > 0.003f) norm = sqrtf(0.003f / nv);
                }
                for (int idx = tid; idx < C * NC; idx += NT) {
                    const int c = idx / NC, i = idx - c * NC;                      // packed index i = samples 2i, 2i+1
                    cf v = mk(0.0f, 0.0f);
                    if (i < NC / 2) {
                        v = cscale(res[c * Sh::NCP + i], inv_nc);
                        if (td_constrain) v = cscale(v, norm);                    // gsc_aic.py:93 / gsc_bm.py:94
                        if (p.kind == FDAF_BM && td_constrain) {                  // gsc_bm.py:48-59,97-108
                            float* vv = &v.x;
                            for (int q = 0; q < 2; ++q) {
                                const int s = 2 * i + q, dq = s > N / 4 ? s - N / 4 : N / 4 - s;
                                const float up = dq == 0 ? 0.9f : dq == 1 ? 0.3f : dq == 2 ? 0.05f : 0.001f;
                                vv[q] = fminf_(fmaxf_(vv[q], -0.001f), up);
                            }
                        }
                        if (want_w) {                                             // self.w = irfft(W)[:filter_len]   :237-238
                            p.w_out[((long long)b * HOP + 2 * i) * C + c] = v.x;
                            p.w_out[((long long)b * HOP + 2 * i + 1) * C + c] = v.y;
                        }
                        if (p.trunc >= 0) {                                       // :240-244
                            float* vv = &v.x;
                            for (int q = 0; q < 2; ++q) {
                                const int s = 2 * i + q;
                                // w_shift[:ft] = 0; w_shift[-ft:] = 0 — a slice [-0:] is the whole array
                                if (s < p.trunc || s >= HOP - p.trunc || p.trunc == 0) vv[q] = 0.0f;
                                else if (use_norm) vv[q] *= norm;                 // gsc_aic.py:106
                            }
                        }
                    }
                    fa[c * Sh::NCP + i] = v;
                }
            });
            if (!rewrite_W) continue;
            fft<-1>(ex, sh, C);
            ex.phase([&](int tid, Rg& r) {
                const int k = tid;
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) r.W[c] = fdaf_split<NC>(res + c * Sh::NCP, k, sh.tb.tw);
                if (k == 0)
                    for (int c = 0; c < C; ++c) {
                        const cf F0 = res[c * Sh::NCP];
                        sh.nyqW[c] = mk(F0.x - F0.y, 0.0f);
                    }
            });
        }

        ex.phase([&](int tid, Rg& r) {
            for (int i = tid; i < C * HOP; i += NT) xog[i] = sh.xold[i / HOP][i % HOP];
            for (int i = tid; i < HOP / 2; i += NT) dog[i] = sh.dold[i];
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) Wg[c * K + tid] = r.W[c];
            Pg[tid] = r.P;
            if (two_path) {
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) Fg[c * K + tid] = r.F[c];
            }
            if (tid == 0) {
                for (int c = 0; c < C; ++c) Wg[c * K + NC] = sh.nyqW[c];
                if (two_path) for (int c = 0; c < C; ++c) Fg[c * K + NC] = sh.nyqF[c];
                Pg[NC] = sh.nyqP;
            }
        });
    }
};

}  // namespace ds
