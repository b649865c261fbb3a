// ds_front.hpp — the front end of the SubbandGSC chain as ONE kernel per utterance for short calls (the realtime regime):
//   DC notch per channel (feature.py:32-49)  ->  TimeAlignment FIR bank (fixedbeamformer.py:13-93)  ->  channel mean = fixed beamformer
//   (SubbandGSC.py:143,206)  ->  analysis of the M aligned channels AND of the fixed output (transform.py:430-453)
// — four launches of the chain (notch, FIR, M-channel analysis, single-channel analysis) and their intermediate round trips through
// HBM (the notched and the aligned samples) in one: at one block per call those kernels are latency-bound (notch 15 us, FIR 30 us, the
// analyses 24 + 16 us for 0.2 GB, profiles/r02c/cfg5_timeline.txt).  One workgroup = one utterance, NT = NFFT / 2 threads:
//   per block: raw hop -> LDS; lanes 0 .. M-1 run the notch recursions (serial in time: 16-sample register chunks) into the FIR window
//   [L - 1 history | hop]; thread i forms output sample i of every channel (taps ascending, the accumulation order of td_fir) and the
//   channel mean straight into the transform's input buffer; the packed real FFT of M + 1 channels is Engine's plan; the split writes
//   D [T][K][M] and F [T][K].
// The same arithmetic in the same order as ds_dcnotch_kernel / ds_fir_kernel / StftEngine: the chain's results do not depend on which
// front end ran (tests).  State: notch memories, FIR history (ping-pong pair, parity device-resident), the two transforms' input tails.
#pragma once
#include "ds_core.hpp"

namespace ds {

constexpr int FRONT_LMAX = 128;      // FIR taps the window buffer holds (the TimeAlignment bank has 84)

struct FrontParams {
    int B, T, L;                 // utterances, blocks in this call, FIR taps
    const float* x;              // input [B][M][n] with element strides
    long long x_bstride, x_cstride;
    float* mem;                  // notch memories [B][M][2]
    float radius;
    const float* coef;           // FIR [L][M]
    const float* cache_in;       // FIR history [B][L-1][M] (ping-pong pair; dev_parity odd = roles swapped)
    float* cache_out;
    const int* dev_parity;
    float* tail_d;               // analysis tail of the M aligned channels [B][M][hop]   (Transform.previous_input of SubbandGSC.transform)
    float* tail_f;               // analysis tail of the fixed output [B][hop]             (bm[m].transform_x)
    const vec4* tables;
    float* D;                    // complex [B][T][K][M]
    float* F;                    // complex [B][T][K]
    float* fixed;                // [B][n] fixed beamformer output (always written: the chain's one-block delay carries it)
    float* xa;                   // [B][M][n] aligned channels, or null
    TickArgs tick;
};

template <int NFFT, int M> struct FrontEngine {
    static constexpr int MC = M + 1, N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2, NT = NC;
    static constexpr int NCP = NC + NC / 4;
    static constexpr bool WAVE_FFT = NC == 256;
    static constexpr bool FWD_FINAL_IS_FB = (NC != 512);
    static_assert(FRONT_LMAX - 1 <= HOP && FRONT_LMAX % 4 == 0, "the FIR history must fit one hop");
    static constexpr int HB = FRONT_LMAX;                        // window entry of sample 0 of the current hop (a multiple of 4)
    static constexpr int RL = (HB + HOP) / 4;                    // columns of a phase row
    struct Sh {
        static constexpr int NCP = NFFT / 2 + NFFT / 8;
        alignas(16) float xbuf[MC][N];            // [old hop | new hop] of the aligned channels and (last) the fixed output; roles swap every block
        cf fa[MC][NCP];
        cf fb[MC][NCP];
        alignas(16) Tables<NFFT> tb;
        // FIR window per channel, entries e = HB - (L - 1) .. HB + HOP - 1 (history, then the notched hop: sample s at e = HB + s), stored
        // phase-split — entry e at [e & 3][e >> 2] — so that the lanes of a wave, four consecutive outputs each, read any tap without bank
        // conflicts (their entries differ by multiples of 4: same phase row, consecutive columns)
        float win[M][4][RL];
        float coef[M][FRONT_LMAX];
    };
    static DS_HD int at(int e) { return (e & 3) * RL + (e >> 2); }
    struct Rg { int unused; };

    template <class Exec> static DS_HD void run(Exec& ex, const FrontParams& p, int b, Sh& sh) {
        const int L = p.L, H = L - 1;
        const float* xg = p.x + (long long)b * p.x_bstride;
        float* memg = p.mem + (long long)b * M * 2;
        const bool swapped = p.dev_parity != nullptr && (p.dev_parity[0] & 1);
        const float* cin = (swapped ? p.cache_out : p.cache_in) + (long long)b * H * M;
        float* cout = (swapped ? const_cast<float*>(p.cache_in) : p.cache_out) + (long long)b * H * M;
        float* tdg = p.tail_d + (long long)b * M * HOP;
        float* tfg = p.tail_f + (long long)b * HOP;
        const long long n = (long long)p.T * HOP;
        cf* Dg = reinterpret_cast<cf*>(p.D) + (long long)b * p.T * K * M;
        cf* Fg = reinterpret_cast<cf*>(p.F) + (long long)b * p.T * K;
        const float r = p.radius;
        const float den2 = fma_(r, r, 0.7f * (1.0f - r) * (1.0f - r));
        cf* fa = &sh.fa[0][0];
        cf* fb = &sh.fb[0][0];
        int old_half = 0;

        ex.phase([&](int tid, Rg&) {
            vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
            for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
            for (int i = tid; i < M * HOP; i += NT) sh.xbuf[i / HOP][i % HOP] = tdg[i];           // the transforms' carried tails -> old halves
            for (int i = tid; i < HOP; i += NT) sh.xbuf[M][i] = tfg[i];
            for (int i = tid; i < M * H; i += NT) { const int j = i / M, m = i - j * M; (&sh.win[m][0][0])[at(HB - H + j)] = cin[i]; }   // history [L-1][M], oldest first
            for (int i = tid; i < M * L; i += NT) { const int j = i / M, m = i - j * M; sh.coef[m][j] = p.coef[i]; }
        });

        for (int t = 0; t < p.T; ++t) {
            const int new_half = old_half ^ 1;
            // ---- raw hop of every channel into the window behind the history
            ex.phase([&](int tid, Rg&) {
                for (int i = tid; i < M * HOP; i += NT) {
                    const int m = i / HOP, s = i - m * HOP;
                    (&sh.win[m][0][0])[at(HB + s)] = xg[(long long)m * p.x_cstride + (long long)t * HOP + s];
                }
            });
            // ---- DC notch in place, one lane per channel (the recursion is serial in time)
            ex.phase([&](int tid, Rg&) {
                if (tid < M) {
                    float m0 = memg[2 * tid], m1 = memg[2 * tid + 1];
                    float* wm = &sh.win[tid][0][0];
                    auto step = [&](float vin) {
                        const float vout = m0 + vin;
                        m0 = m1 + 2.0f * (-vin + r * vout);
                        m1 = vin - den2 * vout;
                        return r * vout;
                    };
                    for (int i = 0; i < HOP; i += 16) {                          // 16 samples = 4 columns of every phase row
                        float a[16];
#pragma unroll
                        for (int q = 0; q < 16; ++q) a[q] = wm[(q & 3) * RL + ((HB + i) >> 2) + (q >> 2)];
#pragma unroll
                        for (int q = 0; q < 16; ++q) a[q] = step(a[q]);
#pragma unroll
                        for (int q = 0; q < 16; ++q) wm[(q & 3) * RL + ((HB + i) >> 2) + (q >> 2)] = a[q];
                    }
                    memg[2 * tid] = m0; memg[2 * tid + 1] = m1;
                }
            });
            // ---- FIR bank: work item = (channel m, four consecutive outputs 4q .. 4q+3), a wavefront per channel (channels 0 .. 3, then the
            // rest); per tap ONE window read per lane (the four outputs' operands slide through registers) and one broadcast coefficient
            // read; every output accumulates its taps in ascending order (td_fir's order)
            ex.phase([&](int tid, Rg&) {
                for (int item = tid; item < M * (HOP / 4); item += NT) {
                    const int m = item / (HOP / 4), q = item - m * (HOP / 4);
                    const float* wm = &sh.win[m][0][0];
                    const float* c = sh.coef[m];
                    float v0 = wm[0 * RL + HB / 4 + q], v1 = wm[1 * RL + HB / 4 + q], v2 = wm[2 * RL + HB / 4 + q], v3 = wm[3 * RL + HB / 4 + q];
                    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                    int j = 0;
                    for (; j + 4 <= L; j += 4) {                                 // taps j .. j+3; entering: v_i = x[4q + i - j]
                        const int col = HB / 4 + q - (j >> 2) - 1;               // column of the entries 4q - j - 1 .. 4q - j - 4 (phases 3, 2, 1, 0)
                        const float n3 = wm[3 * RL + col], n2 = wm[2 * RL + col], n1 = wm[1 * RL + col], n0 = wm[0 * RL + col];
                        const float c0 = c[j], c1 = c[j + 1], c2 = c[j + 2], c3 = c[j + 3];
                        a0 = fma_(c0, v0, a0); a1 = fma_(c0, v1, a1); a2 = fma_(c0, v2, a2); a3 = fma_(c0, v3, a3);
                        a0 = fma_(c1, n3, a0); a1 = fma_(c1, v0, a1); a2 = fma_(c1, v1, a2); a3 = fma_(c1, v2, a3);
                        a0 = fma_(c2, n2, a0); a1 = fma_(c2, n3, a1); a2 = fma_(c2, v0, a2); a3 = fma_(c2, v1, a3);
                        a0 = fma_(c3, n1, a0); a1 = fma_(c3, n2, a1); a2 = fma_(c3, n3, a2); a3 = fma_(c3, v0, a3);
                        v3 = n3; v2 = n2; v1 = n1; v0 = n0;                      // x[4q + i - (j + 4)]
                    }
                    for (; j < L; ++j) {                                         // remaining taps (L not a multiple of 4), one at a time
                        const float cj = c[j];
                        a0 = fma_(cj, v0, a0); a1 = fma_(cj, v1, a1); a2 = fma_(cj, v2, a2); a3 = fma_(cj, v3, a3);
                        const float nx = wm[at(HB + 4 * q - j - 1)];
                        v3 = v2; v2 = v1; v1 = v0; v0 = nx;
                    }
                    float* o = &sh.xbuf[m][new_half * HOP + 4 * q];
                    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
                }
            });
            // ---- channel mean = the fixed beamformer output (channels added in ascending order), the samples out to HBM
            ex.phase([&](int tid, Rg&) {
                for (int s = tid; s < HOP; s += NT) {
                    float acc_mean = 0.0f;
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const float acc = sh.xbuf[m][new_half * HOP + s];
                        if (p.xa) p.xa[((long long)b * M + m) * n + (long long)t * HOP + s] = acc;
                        acc_mean += acc;
                    }
                    const float fx = acc_mean / (float)M;
                    sh.xbuf[M][new_half * HOP + s] = fx;
                    p.fixed[(long long)b * n + (long long)t * HOP + s] = fx;
                }
            });
            // ---- the window's history for the next block: its last L - 1 entries move to the front (L - 1 <= HOP: source and destination
            // do not overlap), then M + 1 packed real transforms (Engine's plan; a channel stays in one wavefront between the 512-point stages)
            auto ph = [&](bool wave_local, auto f) { if (wave_local) ex.phase_wave(f); else ex.phase(f); };
            ph(WAVE_FFT, [&](int tid, Rg&) {
                for (int i = tid; i < M * H; i += NT) {
                    const int m = i / H, e = HB - H + (i - m * H);
                    (&sh.win[m][0][0])[at(e)] = (&sh.win[m][0][0])[at(e + HOP)];
                }
                fft_stage<NFFT, MC, 4, -1, true, 0, 1>(tid, NT, sh, nullptr, fa, 1, old_half, MC);
            });
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, MC, 4, -1, false, 1, 2>(tid, NT, sh, fa, fb, 4, 0, MC); });
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, MC, 4, -1, false, 2, 0>(tid, NT, sh, fb, fa, 16, 0, MC); });
            if (NC == 128) {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, MC, 2, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, MC); });
            } else {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, MC, 4, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, MC); });
                if (NC == 512)
                    ex.phase([&](int tid, Rg&) { fft_stage<NFFT, MC, 2, -1, false, 0, 0>(tid, NT, sh, fb, fa, 256, 0, MC); });
            }
            const cf* Fq = FWD_FINAL_IS_FB ? fb : fa;
            ex.phase([&](int tid, Rg&) {
                const int k = tid, k2 = (NC - k) & (NC - 1);
                const cf w = sh.tb.tw[k];
                cf* dst = Dg + ((long long)t * K + k) * M;
#pragma unroll
                for (int m = 0; m < MC; ++m) {
                    const cf A = Fq[m * NCP + k], Bc = cconj(Fq[m * NCP + k2]);
                    const cf E = cscale(cadd(A, Bc), 0.5f);
                    const cf Dd = csub(A, Bc);
                    const cf O = mk(0.5f * Dd.y, -0.5f * Dd.x);
                    cf Z = cfma(E, w, O);
                    if (k == 0) Z.y = 0.0f;
                    if (m < M) dst[m] = Z; else Fg[(long long)t * K + k] = Z;
                }
                if (k == 0) {                                                       // Nyquist bin
                    cf* dn = Dg + ((long long)t * K + NC) * M;
#pragma unroll
                    for (int m = 0; m < MC; ++m) {
                        const cf F0 = Fq[m * NCP];
                        const cf Zn = mk(F0.x - F0.y, 0.0f);
                        if (m < M) dn[m] = Zn; else Fg[(long long)t * K + NC] = Zn;
                    }
                }
            });
            old_half = new_half;
        }

        ex.phase([&](int tid, Rg&) {
            for (int i = tid; i < M * HOP; i += NT) tdg[i] = sh.xbuf[i / HOP][old_half * HOP + i % HOP];
            for (int i = tid; i < HOP; i += NT) tfg[i] = sh.xbuf[M][old_half * HOP + i];
            for (int i = tid; i < M * H; i += NT) { const int j = i / M, m = i - j * M; cout[i] = (&sh.win[m][0][0])[at(HB - H + j)]; }
        });
    }
};

}  // namespace ds
