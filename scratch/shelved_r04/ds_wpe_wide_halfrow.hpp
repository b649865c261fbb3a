// ds_wpe_wide.hpp — RLS-WPE (Wpe.update, dereverberation/awpe.py:129-192) for wide prediction filters: 16 < C N <= 80, the
// operating point of the reference's maintained use, Wpe(channels=4, filter_len=20, delay=4, num_bands=256, hop_length=64)
// (example/wpe.ipynb cell 2: CN = 80), and of SURVEY 8(d)'s 8-channel x 10-tap sizing of BASELINE config 4.
//
// ONE WORKGROUP PER (utterance, bin), TWO LANES PER ROW.  The inverse correlation matrix P (CN x CN complex, Hermitian) lives in the
// workgroup's registers for all T frames of the call: lane l holds columns (l & 1) HC .. + HC of row l >> 1 (HC = CNP / 2; CNP = CN padded
// to 32 / 64 / 80; 2 CNP lanes = one, two or three wavefronts, the last one part filled).  40 complex words per lane at CN = 80.
// (Round 4's first form held a whole row per lane in ONE wavefront — 200 registers of matrix, two waves per SIMD: every phase of a bin,
// its loads included, ran at the latency of a lone wave, and the kernel stopped at 0.65 of the HBM roofline with the vector pipes 41 %
// busy.  The same matrix over 2.5 x the lanes finishes a bin that much sooner with as many bytes in flight per CU.)
// g = P x is a lane-local dot product against the tap buffer broadcast from LDS and one pair sum; the rank-1 downdate
// P <- (P - g g^H / den) / lambda is lane-local given h = g sqrt(1 / (den lambda)) broadcast from LDS: P / lambda - h_i conj(h_j), four packed
// instructions per element.  Both halves of the matrix are computed; herm_downdate_h() rounds element (i, j) to the exact conjugate of
// (j, i), so P stays Hermitian bit for bit and only its upper triangle is state in HBM (ds_wpe.hpp's block layout, unchanged: packed
// triangle by columns, W, taps, var).  At one frame per call the kernel is HBM-bound on that block (29 KB per bin at CN = 80, once in and
// once out); the redundant half of the arithmetic is vector work the regime has to spare.
//
// The prediction filters W (C x CN) are spread as strips: LANES / Cp lanes per channel (Cp = C rounded up to a power of two), SL taps
// per lane, so that a lane's share of the filter output is a short dot product and its weight update needs one error value.
//
// The packed triangle passes through an LDS tile on its way in and out, the whole of it at once (26 KB at CN = 80: five workgroups per
// CU): 16-byte coalesced pieces on the HBM side — inbound as LDS-DMA (global_load_lds_dwordx4: every piece in flight at once, no
// registers) — and on the register side lane (i, half) reads its half row: P[i][q] above the diagonal as stored, conj(P[q][i]) below it.
//
// Padding: rows / columns / taps CN .. CNP - 1 are zero in registers and LDS and stay zero under every update (g = 0 there), so the
// frame loop has no shape guards; only the state's load and store know CN.  Same statement order whatever T: a call of T frames is
// bit for bit T one-frame calls.  Phases end in workgroup barriers (Exec::phase); tests/emul runs the program serially.
#pragma once
#include "ds_wpe.hpp"

namespace ds {

DS_HD constexpr int wpew_words(int c) { return c * (c + 1) / 2; }      // packed words in front of column c

template <int CNP> struct WpeWideDims {
    static_assert(CNP % 16 == 0 && CNP >= 32 && CNP <= WPEW_CNMAX, "padded size");
    static constexpr int HC = CNP / 2;                  // columns per lane
    static constexpr int LANES = 2 * CNP;               // lanes that hold a half row
    static constexpr int NT = (LANES + 63) & ~63;       // workgroup size (whole wavefronts)
    static constexpr int SLP = 4;                       // strip registers of W: C CN / (LANES / Cp) <= 4 (C = 8: LANES / 8 lanes per channel)
    static constexpr int XP = 2 * CNP;                  // tap buffer / g in LDS: zero beyond CN (a one-channel filter's strips read up to lane LANES - 1)
    static constexpr int TILE = (wpew_words(CNP) + 1) & ~1;
};

template <int CNP, bool GEO = true> struct WpeWideShared {
    typedef WpeWideDims<CNP> D;
    alignas(16) cf tile[D::TILE];
    alignas(16) cf X[2][D::XP + WPE_CMAX];     // tap buffer, double-buffered; [XP + c] = the frame's delayed input of channel c
    alignas(16) cf g[D::XP];                   // g = P x, then h = g sqrt(1 / (den lambda))
    alignas(16) cf red[D::NT];                 // a lane's half of its row's product; then its share of the filter output of its channel
    alignas(16) float dre[D::XP];              // Re(conj(x_i) g_i)
    alignas(16) float p16[16];
    float ks;                                  // kn = h ks
    cf d[WPE_CMAX], err[WPE_CMAX];
    int geo[GEO ? 3 : 1][GEO ? D::NT : 1];     // run-time shapes: a lane's tap source and W strip (src, wc, wi0)
};

template <int CNP> struct WpeWideRegs {
    typedef WpeWideDims<CNP> D;
    cf Pa[D::HC];
    cf W[D::SLP];
    cf xin, din;             // next frame's inputs (lanes < C)
    float var;
    long long io0, ring0;
};

// CT > 0: the channel count as a compile-time constant (strip geometry and the channel loops fold); NTAPS with it
template <int CNP, int CT = 0, int NTAPS = 0> struct WpeWideEngine {
    typedef WpeWideDims<CNP> D;
    typedef WpeWideShared<CNP, CT == 0> Sh;
    typedef WpeWideRegs<CNP> Rg;
    static constexpr int NT = D::NT, LANES = D::LANES, HC = D::HC, SLP = D::SLP, XP = D::XP;
    static_assert(CT * NTAPS <= CNP, "shape");

    template <class Exec> static DS_HD void run(Exec& ex, const WpeParams& p, int blk, Sh& sh) {
        const int C = CT > 0 ? CT : p.C, N = NTAPS > 0 ? NTAPS : p.N, CN = C * N;
        const int SB = wpe_bin_floats(C, N), NPK = wpe_packed(CN);
        const float lam = p.lam, lam_inv = 1.0f / p.lam;
        const int ring_pos = p.dev_ring_pos ? p.dev_ring_pos[0] : p.ring_pos;
        const long long gbin = blk;                                    // one workgroup per (utterance, bin)
        const long long ub = gbin / p.K, kb = gbin - ub * p.K;
        float* const stf = p.state + ub * p.ustride + kb * SB;
        cf* const st = reinterpret_cast<cf*>(stf);
        const long long fstride = (long long)p.K * C;
        // W strips: LPC lanes per channel (of the LANES that hold matrix words), SL taps per lane
        const int Cp = C <= 1 ? 1 : C <= 2 ? 2 : C <= 4 ? 4 : 8, LPC = LANES / Cp, SL = (CN + LPC - 1) / LPC;
        // tap i of the buffer <- tap i - 1 of the same channel, or the channel's new (delayed) frame at its tap 0 (awpe.py:80-102);
        // the compile-time shapes recompute a lane's geometry where it is used (a division by a constant), the run-time shapes keep it
        // in LDS (sh.geo).  src: source word of tap l in the previous buffer (XP + c = channel c's new frame), -1 = no such tap;
        // wc, wi0: channel (or -1) and first tap of the lane's strip of W
        auto src_of = [&](int i) { return i >= CN ? -1 : (i % N == 0 ? XP + i / N : i - 1); };
        auto wc_calc = [&](int l) { if (l >= LANES) return -1; const int c = l / LPC; return c < C ? c : -1; };
        auto src0_of = [&](int l) { return CT > 0 ? src_of(l) : sh.geo[0][l]; };
        auto wc_of = [&](int l) { return CT > 0 ? wc_calc(l) : sh.geo[1][l]; };
        auto wi0_of = [&](int l) { return CT > 0 ? (l % LPC) * SL : sh.geo[2][l]; };
        auto io_at = [&](const Rg& r, int t) { return r.io0 + (long long)t * fstride; };
        auto ring_slot = [&](const Rg& r, int s) { return r.ring0 + (long long)s * fstride; };
        auto delayed = [&](const Rg& r, int t, int c) {             // x_delayed[c] of frame t (ds_wpe.hpp: one load from a selected address)
            const float* src = p.xd;
            long long f = io_at(r, t);
            if (p.ring != nullptr) {
                const bool in_ring = t < p.ring_len;
                src = in_ring ? p.ring : p.d;
                f = in_ring ? ring_slot(r, (ring_pos + t) % p.ring_len) : io_at(r, t - p.ring_len);
            }
            return mk(src[2 * (f + c)], src[2 * (f + c) + 1]);
        };

        // ---- prologue: LDS to zero, geometry, the small parts of the state, the first frame's inputs, the triangle on its way (LDS-DMA:
        // a wavefront's instruction lands 64 consecutive 16-byte pieces behind a uniform base; every piece of the tile is in flight at once —
        // with ordinary loads the copy loop ran load, wait, store piece by piece: two HBM latencies per KiB)
        ex.phase([&](int l, Rg& r) {
            const cf z = mk(0.0f, 0.0f);
#pragma unroll
            for (int s = 0; s < SLP; ++s) r.W[s] = z;
            for (int i = l; i < XP + WPE_CMAX; i += NT) { sh.X[0][i] = z; sh.X[1][i] = z; }
            for (int i = l; i < XP; i += NT) { sh.g[i] = z; sh.dre[i] = 0.0f; }
            sh.red[l] = z;
            const int wc = wc_calc(l), wi0 = (l % LPC) * SL;
            if constexpr (CT == 0) { sh.geo[0][l] = src_of(l); sh.geo[1][l] = wc; sh.geo[2][l] = wi0; }
            if (wc >= 0) {
#pragma unroll
                for (int s = 0; s < SLP; ++s)
                    if (s < SL && wi0 + s < CN) r.W[s] = st[NPK + wc * CN + wi0 + s];
            }
            r.var = stf[2 * (NPK + C * CN + CN)];
            r.io0 = (ub * p.T * p.K + kb) * C;
            r.ring0 = p.ring != nullptr ? (ub * p.ring_len * p.K + kb) * C : 0;
            r.xin = z; r.din = z;
            if (l < C) { r.xin = delayed(r, 0, l); r.din = mk(p.d[2 * (r.io0 + l)], p.d[2 * (r.io0 + l) + 1]); }
            // the packed triangle: words [0, NPK) as 16-byte pieces (two words); a wavefront covers 128 words per instruction
            const int we = NPK & ~1, wv = l >> 6, ln = l & 63;
            for (int w = wv * 128; w < we; w += (NT / 64) * 128)
                if (w + 2 * ln < we) ex.lds_load16(&sh.tile[w], ln, &st[w + 2 * ln]);
            if ((NPK & 1) && l == 0) sh.tile[NPK - 1] = st[NPK - 1];
            ex.lds_load_wait();
        });
        ex.phase([&](int l, Rg& r) {                                  // (after the zero fill) the taps as stored; this lane's half row
            for (int i = l; i < CN; i += NT) sh.X[0][i] = st[NPK + C * CN + i];
            // row i, columns q0 .. q0 + HC: P[i][q] above the diagonal as stored, the conjugate of the row's own column below it;
            // words(q) advances by q + 1 per column
            const int i = l >> 1, q0 = (l & 1) * HC;
            const bool on = l < LANES && i < CN;
            const int rowbase = wpew_words(i);
            int wq = wpew_words(q0);
#pragma unroll
            for (int j = 0; j < HC; ++j) {
                const int q = q0 + j;
                const bool up = i <= q, ok = on && q < CN;
                const cf v = sh.tile[ok ? (up ? wq + i : rowbase + q) : 0];
                r.Pa[j] = mk(ok ? v.x : 0.0f, ok ? (up ? v.y : -v.y) : 0.0f);
                wq += q + 1;
            }
        });

        int cur = 0;
        for (int t = 0; t < p.T; ++t) {
            const int nxt = cur ^ 1;
            // ---- the frame's inputs to LDS; the chain's delay line keeps the call's last ring_len frames
            ex.phase([&](int l, Rg& r) {
                if (l < C) {
                    sh.X[cur][XP + l] = r.xin;
                    sh.d[l] = r.din;
                    if (p.ring != nullptr && t >= p.T - p.ring_len) {
                        const long long f = ring_slot(r, (ring_pos + t) % p.ring_len);
                        p.ring[2 * (f + l)] = r.din.x; p.ring[2 * (f + l) + 1] = r.din.y;
                    }
                }
            });
            // ---- buffer_input (awpe.py:80-102)
            ex.phase([&](int l, Rg&) {
                if (l < CNP) { const int s0 = src0_of(l); if (s0 >= 0) sh.X[nxt][l] = sh.X[cur][s0]; }
            });
            // ---- lane-local products: this lane's half of (P x)_i
            ex.phase([&](int l, Rg& r) {
                const cf* X = sh.X[nxt];
                const int q0 = (l & 1) * HC;
                cf a0 = mk(0.0f, 0.0f), a1 = mk(0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < HC; j += 2) {
                    const vec4 x2 = *reinterpret_cast<const vec4*>(&X[q0 + j]);
                    a0 = cfma(a0, r.Pa[j], mk(x2.x, x2.y));
                    a1 = cfma(a1, r.Pa[j + 1], mk(x2.z, x2.w));
                }
                sh.red[l] = cadd(a0, a1);
                if (t + 1 < p.T && l < C) {                                        // next frame's inputs: in flight behind this frame's arithmetic
                    const long long f1 = io_at(r, t + 1);
                    r.xin = delayed(r, t + 1, l);
                    r.din = mk(p.d[2 * (f1 + l)], p.d[2 * (f1 + l) + 1]);
                }
            });
            // ---- g_i = the two halves; Re(conj(x_i) g_i)
            ex.phase([&](int l, Rg&) {
                if (l < CNP) {
                    const vec4 h2 = *reinterpret_cast<const vec4*>(&sh.red[2 * l]);
                    const cf gi = cadd(mk(h2.x, h2.y), mk(h2.z, h2.w));
                    sh.g[l] = gi;
                    const cf xl = sh.X[nxt][l];
                    sh.dre[l] = fma_(xl.x, gi.x, xl.y * gi.y);
                }
            });
            // ---- the strips' shares of the filter outputs (red is free again); den, first level: sixteen lanes sum CNP / 16 terms each
            ex.phase([&](int l, Rg& r) {
                const cf* X = sh.X[nxt];
                cf o = mk(0.0f, 0.0f);
                const int wi0 = wi0_of(l);
#pragma unroll
                for (int s = 0; s < SLP; ++s)
                    if (s < SL) o = cadd(o, cmulc(X[wi0 + s], r.W[s]));            // conj(W[c][i]) x_i  (awpe.py:158)
                sh.red[l] = o;
                if (l < 16) {
                    float a = 0.0f;
#pragma unroll
                    for (int m = 0; m < CNP / 16; ++m) a += sh.dre[l * (CNP / 16) + m];
                    sh.p16[l] = a;
                }
            });
            // ---- the prior error of every channel (awpe.py:158-161): the first lane of its strips
            ex.phase([&](int l, Rg& r) {
                const int wc = wc_of(l);
                if (wc >= 0 && l == wc * LPC) {
                    cf o = mk(0.0f, 0.0f);
                    for (int m = 0; m < LPC; ++m) o = cadd(o, sh.red[l + m]);
                    const cf e = csub(sh.d[wc], o);
                    sh.err[wc] = e;
                    const long long f = io_at(r, t);
                    p.err[2 * (f + wc)] = e.x; p.err[2 * (f + wc) + 1] = e.y;
                    if (p.err0 != nullptr && wc == 0) {
                        const long long f0 = (ub * p.T + t) * p.K + kb;
                        p.err0[2 * f0] = e.x; p.err0[2 * f0 + 1] = e.y;
                    }
                }
            });
            // ---- gain (awpe.py:163-180): var, den; g is rescaled in place to h = g sqrt(1 / (den lambda)), so that the downdate below is
            // P / lambda - h_i conj(h_j): four packed instructions per element instead of five
            ex.phase([&](int l, Rg& r) {
                float dpow = 0.0f;
                for (int c = 0; c < C; ++c) dpow += cabs2(sh.d[c]);
                r.var = fma_(0.98f, r.var, (float)(1.0 - 0.98) * (dpow / (float)C));
                float den = lam * r.var;
#pragma unroll
                for (int m = 0; m < 16; m += 4) {
                    const vec4 s4 = *reinterpret_cast<const vec4*>(&sh.p16[m]);
                    den += s4.x; den += s4.y; den += s4.z; den += s4.w;
                }
                // digital silence from the first frame on (var = 0, x = 0): the reference's gain is 0 / 0; it is 0 here (ds_wpe.hpp).  den > 0
                // otherwise: lambda var >= 0 and x^H P x >= 0
                const float dinv = den > 0.0f ? 1.0f / den : 0.0f;
                const float hs = sqrt_(dinv * lam_inv);
                if (l < CNP) sh.g[l] = cscale(sh.g[l], hs);
                if (l == 0) sh.ks = hs > 0.0f ? dinv / hs : 0.0f;              // kn = g / den = h (dinv / hs)
            });
            // ---- P and W updates (awpe.py:181-189)
            ex.phase([&](int l, Rg& r) {
                const int i = l >> 1, q0 = (l & 1) * HC;
                const cf hi = sh.g[i];                                        // (i < NT / 2 <= XP: rows beyond CN read the zero padding)
#pragma unroll
                for (int j = 0; j < HC; j += 2) {
                    const vec4 h2 = *reinterpret_cast<const vec4*>(&sh.g[q0 + j]);
                    r.Pa[j] = herm_downdate_h(r.Pa[j], hi, mk(h2.x, h2.y), lam_inv);
                    r.Pa[j + 1] = herm_downdate_h(r.Pa[j + 1], hi, mk(h2.z, h2.w), lam_inv);
                }
                const int wc = wc_of(l);
                if (wc >= 0) {
                    const cf e = sh.err[wc];
                    const float ks = sh.ks;
                    const int wi0 = wi0_of(l);
#pragma unroll
                    for (int s = 0; s < SLP; ++s)
                        if (s < SL) r.W[s] = cadd(r.W[s], cmulc(cscale(sh.g[wi0 + s], ks), e));     // W_c += conj(err_c) kn  (awpe.py:188-189)
                }
            });
            cur = nxt;
        }

        // ---- epilogue: the small parts; the upper triangle through the tile
        ex.phase([&](int l, Rg& r) {
            const int wc = wc_of(l), wi0 = wi0_of(l);
            if (wc >= 0) {
#pragma unroll
                for (int s = 0; s < SLP; ++s)
                    if (s < SL && wi0 + s < CN) st[NPK + wc * CN + wi0 + s] = r.W[s];
            }
            for (int i = l; i < CN; i += NT) st[NPK + C * CN + i] = sh.X[cur][i];
            if (l == 0) stf[2 * (NPK + C * CN + CN)] = r.var;
            const int i = l >> 1, q0 = (l & 1) * HC;
            const bool on = l < LANES && i < CN;
            int wq = wpew_words(q0);
#pragma unroll
            for (int j = 0; j < HC; ++j) {
                const int q = q0 + j;
                if (on && i <= q && q < CN) sh.tile[wq + i] = r.Pa[j];
                wq += q + 1;
            }
        });
        ex.phase([&](int l, Rg&) {
            const int we = NPK & ~1;
            for (int w = 2 * l; w < we; w += 2 * NT) *reinterpret_cast<vec4*>(&st[w]) = *reinterpret_cast<const vec4*>(&sh.tile[w]);
            if ((NPK & 1) && l == 0) st[NPK - 1] = sh.tile[NPK - 1];
        });
    }
};

}  // namespace ds
