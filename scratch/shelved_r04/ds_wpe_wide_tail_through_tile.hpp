// ds_wpe_wide.hpp — RLS-WPE (Wpe.update, dereverberation/awpe.py:129-192) for wide prediction filters: 16 < C N <= 80, the
// operating point of the reference's maintained use, Wpe(channels=4, filter_len=20, delay=4, num_bands=256, hop_length=64)
// (example/wpe.ipynb cell 2: CN = 80), and of SURVEY 8(d)'s 8-channel x 10-tap sizing of BASELINE config 4.
//
// ONE WAVEFRONT PER (utterance, bin).  The inverse correlation matrix P (CN x CN complex, Hermitian) lives in the wavefront's
// registers for all T frames of the call:
//   lane l < 64 holds row l of P (CNP complex words; CNP = CN padded to 32 / 64 / 80);
//   rows 64 .. 79 (CN > 64) are held four lanes to a row: lane l has columns (l & 3) QW .. + QW of row 64 + (l >> 2), QW = CNP / 4.
// g = P x is then a lane-local dot product against the tap buffer broadcast from LDS (plus one quad sum for the split rows), and the
// rank-1 downdate P <- (P - g g^H / den) / lambda is lane-local given g broadcast from LDS.  Both halves of the matrix are computed;
// herm_downdate() rounds element (i, j) to the exact conjugate of (j, i), so P stays Hermitian bit for bit and only its upper
// triangle is state in HBM (ds_wpe.hpp's block layout, unchanged: packed triangle by columns, W, taps, var).  At one frame per call
// the kernel is HBM-bound on that block (29 KB per bin at CN = 80, once in and once out); the redundant half of the arithmetic is
// vector work the regime has to spare.
//
// The prediction filters W (C x CN) are spread as strips: 64 / Cp lanes per channel (Cp = C rounded up to a power of two), SL taps per
// lane, so that a lane's share of the filter output is a short dot product and its weight update needs one error value.
//
// The packed triangle passes through an LDS tile on its way in and out (coalesced 16-byte pieces on the HBM side; on the register side
// lane i reads its row: P[i][q] above the diagonal as stored, conj(P[q][i]) below it), in NCH column chunks so that the tile stays
// small (two chunks at CNP = 80: 13 KB, eight wavefronts per CU).
//
// Padding: rows / columns / taps CN .. CNP - 1 are zero in registers and LDS and stay zero under every update (g = 0 there), so the
// frame loop has no shape guards; only the state's load and store know CN.  Same statement order whatever T: a call of T frames is
// bit for bit T one-frame calls.  Every LDS hand-off is inside the wavefront (Exec::phase_wave); tests/emul runs the program serially.
#pragma once
#include <type_traits>
#include "ds_wpe.hpp"

namespace ds {

constexpr int WPEW_NT = 64;

DS_HD constexpr int wpew_words(int c) { return c * (c + 1) / 2; }      // packed words in front of column c

// TAILW > 0 (the compile-time shapes): the words behind the triangle — the gap to the next line, W, the taps, (var, pad) — travel with the
// LAST chunk through the tile as well, so that every byte of the block moves as whole-line, streamed pieces; the chunk boundaries balance
// triangle + tail
template <int CNP, int NCH, int TAILW = 0> struct WpeWideDims {
    static_assert(CNP % 16 == 0 && CNP >= 32 && CNP <= WPEW_CNMAX, "padded size");
    static constexpr int RL = CNP < 64 ? CNP : 64;      // rows held one per lane
    static constexpr int XR = CNP - RL;                 // rows held four lanes to a row
    static constexpr int QW = XR > 0 ? CNP / 4 : 1;     // columns per lane of such a row
    static constexpr int SLP = (CNP + 7) / 8;           // strip registers of W (C = 8: eight lanes per channel)
    static constexpr int XP = CNP + 64;                 // tap buffer / g in LDS: zero beyond CN (strips read past the end)
    static constexpr int NPK = wpew_words(CNP);
    static_assert(XR * 4 <= 64, "split rows");
    // chunk h = columns [col0(h), col0(h + 1)): boundaries where the packed word count is even (c % 4 in {0, 3}), so that every
    // chunk starts on a 16-byte boundary of the block
    static constexpr int col0(int h) {
        if (h <= 0) return 0;
        if (h >= NCH) return CNP;
        int c = 0;
        while (wpew_words(c + 1) <= (long long)h * (NPK + TAILW) / NCH) ++c;
        while (c > 0 && !(c % 4 == 0 || c % 4 == 3)) --c;
        return c;
    }
    static constexpr int tile_words() {
        int m = 0;
        for (int h = 0; h < NCH; ++h) { const int w = wpew_words(col0(h + 1)) - wpew_words(col0(h)) + (h == NCH - 1 ? TAILW : 0); if (w > m) m = w; }
        return (m + 1) & ~1;
    }
    static constexpr int TILE = tile_words();
};

template <int CNP, int NCH, bool GEO = true, int TAILW = 0> struct WpeWideShared {
    typedef WpeWideDims<CNP, NCH, TAILW> D;
    // spare: a word per lane, where a lane stores what is not part of the packed triangle.
    alignas(16) cf tile[D::TILE];
    alignas(16) cf spare[WPEW_NT];
    alignas(16) cf X[2][D::XP + WPE_CMAX];     // tap buffer, double-buffered; [XP + c] = the frame's delayed input of channel c
    alignas(16) cf g[D::XP];                   // g = P x
    alignas(16) cf red[WPEW_NT];               // a lane's share of the filter output of its channel
    alignas(16) cf q[WPEW_NT];                 // a lane's share of g for its split row
    alignas(16) float dre[D::XP];              // Re(conj(x_i) g_i)
    alignas(16) float p16[16];
    float ks;                                  // kn = h ks
    cf d[WPE_CMAX], err[WPE_CMAX];
    int geo[GEO ? 4 : 1][GEO ? WPEW_NT : 1];   // run-time shapes: a lane's tap sources and W strip (src0, src1, wc, wi0)
};

template <int CNP, int NCH> struct WpeWideRegs {
    typedef WpeWideDims<CNP, NCH, 0> D;
    cf Pa[CNP];
    cf Pb[D::QW];
    cf W[D::SLP];
    cf xin, din;             // next frame's inputs (lanes < C)
    float var;
    long long io0, ring0;
};

// CT > 0: the channel count as a compile-time constant (strip geometry and the channel loops fold); NTAPS with it
// (the chunks through two tile buffers as a load pipeline — everything of the prologue and the first two of four chunks in flight before the
// first wait — was built and measured: no gain, profiles/r04a/wpe_wide_pipeline_ab.txt; the kernel is bound by its two waves per SIMD)
template <int CNP, int NCH, int CT = 0, int NTAPS = 0> struct WpeWideEngine {
    // the compile-time shapes move the whole block through the tile: words [NPK, x0 + CN + 2) of wpe_layout() ride with the last chunk
    static constexpr bool TAIL = CT > 0;
    static constexpr int TAILW = TAIL ? wpe_layout(CT, NTAPS).x0 + CT * NTAPS + 2 - wpe_packed(CT * NTAPS) : 0;
    typedef WpeWideDims<CNP, NCH, TAILW> D;
    typedef WpeWideShared<CNP, NCH, CT == 0, TAILW> Sh;
    typedef WpeWideRegs<CNP, NCH> Rg;
    static constexpr int NT = WPEW_NT, RL = D::RL, XR = D::XR, QW = D::QW, SLP = D::SLP, XP = D::XP;
    static_assert(CT * NTAPS <= CNP, "shape");

    template <class Exec> static DS_HD void run(Exec& ex, const WpeParams& p, int blk, Sh& sh) {
        const int C = CT > 0 ? CT : p.C, N = NTAPS > 0 ? NTAPS : p.N, CN = C * N;
        const WpeLayout Lb = wpe_layout(C, N);
        const int SB = Lb.floats, NPK = wpe_packed(CN);
        const float lam = p.lam, lam_inv = 1.0f / p.lam;
        const int ring_pos = p.dev_ring_pos ? p.dev_ring_pos[0] : p.ring_pos;
        const long long gbin = blk;                                    // one workgroup (= one wavefront) per (utterance, bin)
        const long long ub = gbin / p.K, kb = gbin - ub * p.K;
        float* const stf = p.state + ub * p.ustride + kb * SB;
        cf* const st = reinterpret_cast<cf*>(stf);
        const long long fstride = (long long)p.K * C;
        const int Cp = C <= 1 ? 1 : C <= 2 ? 2 : C <= 4 ? 4 : 8, LPC = NT / Cp, SL = (CN + LPC - 1) / LPC;
        // tap i of the buffer <- tap i - 1 of the same channel, or the channel's new (delayed) frame at its tap 0 (awpe.py:80-102);
        // the compile-time shapes recompute a lane's geometry where it is used (a division by a constant), the run-time shapes keep it
        auto src_of = [&](int i) { return i >= CN ? -1 : (i % N == 0 ? XP + i / N : i - 1); };
        // (in LDS: sh.geo).  src: source word of tap l / tap 64 + l in the previous buffer (XP + c = channel c's new frame), -1 = no such tap;
        // wc, wi0: channel (or -1) and first tap of the lane's strip of W
        auto src0_of = [&](int l, const Rg&) { return CT > 0 ? src_of(l) : sh.geo[0][l]; };
        auto src1_of = [&](int l, const Rg&) { return CT > 0 ? src_of(64 + l) : sh.geo[1][l]; };
        // (ALLW: every lane has a strip — C a power of two and LPC SL == CN: no lane predicate around the strip's loads, whose results a
        // predicate would make the compiler wait for on the spot)
        constexpr bool ALLW = CT > 0 && (CT & (CT - 1)) == 0 && ((CT * NTAPS) % (NT / (CT > 0 ? CT : 1))) == 0;
        auto wc_of = [&](int l, const Rg&) { if (ALLW) return l / LPC; if (CT > 0) { const int c = l / LPC; return c < C ? c : -1; } return sh.geo[2][l]; };
        auto wi0_of = [&](int l, const Rg&) { return CT > 0 ? (l % LPC) * SL : sh.geo[3][l]; };
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
        // global <-> tile, 16-byte pieces (two packed words), consecutive lanes on consecutive pieces; words [w0, w1), w0 even
        // (inbound as LDS-DMA: the 13 .. 26 one-KiB pieces of a chunk are all in flight at once and take no registers — with ordinary loads
        // the copy loop ran load, wait, store piece by piece: two HBM latencies per KiB, 39 us per bin at one frame per call)
        auto tile_in = [&](int tid, int w0, int w1) {
            const int we = w1 & ~1;
            for (int w = w0; w < we; w += 2 * NT)
                if (w + 2 * tid < we) ex.lds_load16(&sh.tile[w - w0], tid, &st[w + 2 * tid]);
            if ((w1 & 1) && tid == 0) sh.tile[w1 - 1 - w0] = st[w1 - 1];
            ex.lds_load_wait();
        };
        // (the tile's traffic is non-temporal in both directions — one instruction covers whole 128-byte lines, streamed once per launch:
        // +9 % at one hop per call.  The W strips and taps are 8-byte pieces at a lane stride: as non-temporal stores those cost 24 % at
        // 8 channels, so they stay ordinary; profiles/r04a/wpe_nt_ab.txt)
        auto tile_out = [&](int tid, int w0, int w1) {
            const int we = w1 & ~1;
            for (int w = w0 + 2 * tid; w < we; w += 2 * NT) store_state(reinterpret_cast<vec4*>(&st[w]), *reinterpret_cast<const vec4*>(&sh.tile[w - w0]));
            if ((w1 & 1) && tid == 0) st[w1 - 1] = sh.tile[w1 - 1 - w0];
        };

        // ---- prologue: registers and LDS to zero, geometry, first frame's inputs, the small parts of the state
        ex.phase_wave([&](int l, Rg& r) {
            const cf z = mk(0.0f, 0.0f);
            if constexpr (CT * NTAPS != CNP) {                          // exact shapes: every word of every lane's rows is loaded below
#pragma unroll
                for (int q = 0; q < CNP; ++q) r.Pa[q] = z;
#pragma unroll
                for (int j = 0; j < QW; ++j) r.Pb[j] = z;
            }
#pragma unroll
            for (int s = 0; s < SLP; ++s) r.W[s] = z;
            for (int i = l; i < XP + WPE_CMAX; i += NT) { sh.X[0][i] = z; sh.X[1][i] = z; }
            for (int i = l; i < XP; i += NT) { sh.g[i] = z; sh.dre[i] = 0.0f; }
            sh.red[l] = z; sh.q[l] = z;
            const int c = l / LPC, sub = l - c * LPC;
            const int wc = (ALLW || c < C) ? c : -1, wi0 = sub * SL;
            if constexpr (CT == 0) { sh.geo[0][l] = src_of(l); sh.geo[1][l] = XR > 0 ? src_of(64 + l) : -1; sh.geo[2][l] = wc; sh.geo[3][l] = wi0; }
            if constexpr (!TAIL) {
                if (ALLW || wc >= 0) {
#pragma unroll
                    for (int s = 0; s < SLP; ++s)
                        if (s < SL && (ALLW || wi0 + s < CN)) r.W[s] = st[Lb.w0 + wc * CN + wi0 + s];
                }
                r.var = stf[Lb.var_f];
            } else r.var = 0.0f;                                          // (W, the taps and var arrive with the last chunk of the tile)
            r.io0 = (ub * p.T * p.K + kb) * C;
            r.ring0 = p.ring != nullptr ? (ub * p.ring_len * p.K + kb) * C : 0;
            r.xin = z; r.din = z;
            if (l < C) { r.xin = delayed(r, 0, l); r.din = mk(p.d[2 * (r.io0 + l)], p.d[2 * (r.io0 + l) + 1]); }
        });
        if constexpr (!TAIL)
            ex.phase_wave([&](int l, Rg&) {                           // (after the zero fill: the taps as stored)
                for (int i = l; i < CN; i += NT) sh.X[0][i] = st[Lb.x0 + i];
            });
        // ---- the packed triangle of P, chunk by chunk: block -> tile -> rows
        auto load_chunk = [&](auto hc) {
            constexpr int H = decltype(hc)::value, c0 = D::col0(H), c1r = D::col0(H + 1), w0 = wpew_words(c0);
            if (c0 >= CN) return;
            const int c1 = c1r < CN ? c1r : CN;                         // columns [c0, c1) of the CN the state has
            constexpr bool WITH_TAIL = TAIL && H == NCH - 1;            // ... and, behind the last chunk, W, the taps and var
            const int w1 = WITH_TAIL ? Lb.x0 + CN + 2 : wpew_words(c1);
            ex.phase_wave([&](int l, Rg&) { tile_in(l, w0, w1); });
            // a lane's predicate does not depend on the column: for the chunk's own columns q every row i < c1 takes exactly one word (its
            // stored (i, q) when i <= q, the conjugate of its own column's (q, i) otherwise: then c0 <= q < i < c1), and for the columns in
            // front of the chunk the rows inside it take the conjugates of their column — two straight-line loops under one lane mask each
            // (as one loop with the predicate evaluated per column this was 600 basic blocks of scalar control)
            ex.phase_wave([&](int l, Rg& r) {
                if constexpr (WITH_TAIL) {                              // the lane's strip of W, the taps, var: out of the tile
                    const int wc = wc_of(l, r), wi0 = wi0_of(l, r);
                    if (ALLW || wc >= 0) {
#pragma unroll
                        for (int s = 0; s < SLP; ++s)
                            if (s < SL && (ALLW || wi0 + s < CN)) r.W[s] = sh.tile[Lb.w0 - w0 + wc * CN + wi0 + s];
                    }
                    for (int i2 = l; i2 < CN; i2 += NT) sh.X[0][i2] = sh.tile[Lb.x0 - w0 + i2];
                    r.var = sh.tile[Lb.x0 + CN - w0].x;
                }
                const int i = l;                                        // row l
                const int lowbase = wpew_words(i) - w0;
                if (i < c1) {
#pragma unroll
                    for (int q = c0; q < c1r; ++q)
                        if (q < c1) {                                   // (uniform; folds for the compile-time shapes)
                            const bool up = i <= q;
                            const cf v = sh.tile[up ? wpew_words(q) - w0 + i : lowbase + q];
                            r.Pa[q] = mk(v.x, up ? v.y : -v.y);
                        }
                    if (i >= c0) {
#pragma unroll
                        for (int q = 0; q < c0; ++q) r.Pa[q] = cconj(sh.tile[lowbase + q]);
                    }
                }
                if constexpr (XR > 0 && c1r > 64) {                     // the split rows 64 .. : all in the chunks that reach beyond column 64
                    const int e = 64 + (l >> 2), cq0 = (l & 3) * QW;
                    const int ebase = wpew_words(e) - w0;
                    if constexpr (c0 <= 64 && c1r == CNP) {             // ... normally one chunk: a split row's columns beyond it (stored) and in front of it (its own column) are both here
                        if (e < CN) {
#pragma unroll
                            for (int j = 0; j < QW; ++j) {
                                const int cq = cq0 + j;
                                if (cq < CN) {
                                    const bool up = e <= cq;
                                    const cf v = sh.tile[up ? wpew_words(cq) - w0 + e : ebase + cq];
                                    r.Pb[j] = mk(v.x, up ? v.y : -v.y);
                                }
                            }
                        }
                    } else {
                        const bool ein = e >= c0 && e < c1;
#pragma unroll
                        for (int j = 0; j < QW; ++j) {
                            const int cq = cq0 + j;
                            const bool up = e <= cq;
                            const bool ok = (up ? (cq >= c0 && cq < c1) : (ein && cq < CN)) && e < CN;
                            if (ok) {
                                const cf v = sh.tile[up ? wpew_words(cq) - w0 + e : ebase + cq];
                                r.Pb[j] = mk(v.x, up ? v.y : -v.y);
                            }
                        }
                    }
                }
            });
        };
        load_chunk(std::integral_constant<int, 0>());
        if constexpr (NCH > 1) load_chunk(std::integral_constant<int, 1>());
        if constexpr (NCH > 2) load_chunk(std::integral_constant<int, 2>());
        if constexpr (NCH > 3) load_chunk(std::integral_constant<int, 3>());

        int cur = 0;
        for (int t = 0; t < p.T; ++t) {
            const int nxt = cur ^ 1;
            // ---- the frame's inputs to LDS; the chain's delay line keeps the call's last ring_len frames
            ex.phase_wave([&](int l, Rg& r) {
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
            ex.phase_wave([&](int l, Rg& r) {
                const int s0 = src0_of(l, r);
                if (s0 >= 0) sh.X[nxt][l] = sh.X[cur][s0];
                if constexpr (XR > 0) { const int s1 = src1_of(l, r); if (s1 >= 0) sh.X[nxt][64 + l] = sh.X[cur][s1]; }
            });
            // ---- lane-local products: g_l = (P x)_l, the split rows' shares, the lane's share of its channel's filter output
            ex.phase_wave([&](int l, Rg& r) {
                const cf* X = sh.X[nxt];
                // four partial sums (taps j mod 4): a complex multiply-add is two dependent packed instructions, and with two waves per SIMD
                // the chain of a partial sum is what the row product waits on
                cf a0 = mk(0.0f, 0.0f), a1 = mk(0.0f, 0.0f), a2 = mk(0.0f, 0.0f), a3 = mk(0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < CNP; j += 4) {
                    const vec4 x2 = *reinterpret_cast<const vec4*>(&X[j]), x3 = *reinterpret_cast<const vec4*>(&X[j + 2]);
                    a0 = cfma(a0, r.Pa[j], mk(x2.x, x2.y));
                    a1 = cfma(a1, r.Pa[j + 1], mk(x2.z, x2.w));
                    a2 = cfma(a2, r.Pa[j + 2], mk(x3.x, x3.y));
                    a3 = cfma(a3, r.Pa[j + 3], mk(x3.z, x3.w));
                }
                sh.g[l] = cadd(cadd(a0, a1), cadd(a2, a3));
                if constexpr (XR > 0) {
                    const int cq0 = (l & 3) * QW;
                    cf b0 = mk(0.0f, 0.0f), b1 = mk(0.0f, 0.0f);
#pragma unroll
                    for (int j = 0; j < QW; j += 2) {
                        b0 = cfma(b0, r.Pb[j], X[cq0 + j]);
                        if (j + 1 < QW) b1 = cfma(b1, r.Pb[j + 1], X[cq0 + j + 1]);
                    }
                    sh.q[l] = cadd(b0, b1);
                }
                cf o = mk(0.0f, 0.0f);
                const int wi0 = wi0_of(l, r);
#pragma unroll
                for (int s = 0; s < SLP; ++s)
                    if (s < SL) o = cadd(o, cmulc(X[wi0 + s], r.W[s]));            // conj(W[c][i]) x_i  (awpe.py:158)
                sh.red[l] = o;
                if (t + 1 < p.T && l < C) {                                        // next frame's inputs: in flight behind this frame's arithmetic
                    const long long f1 = io_at(r, t + 1);
                    r.xin = delayed(r, t + 1, l);
                    r.din = mk(p.d[2 * (f1 + l)], p.d[2 * (f1 + l) + 1]);
                }
            });
            // ---- quad sums of the split rows; Re(conj(x_i) g_i); the prior error of the lane's channel (awpe.py:158-161)
            ex.phase_wave([&](int l, Rg& r) {
                const cf* X = sh.X[nxt];
                const cf xl = X[l], gl = sh.g[l];
                sh.dre[l] = fma_(xl.x, gl.x, xl.y * gl.y);
                if constexpr (XR > 0) {
                    if ((l & 3) == 0) {
                        const int e = l >> 2;
                        const vec4 q01 = *reinterpret_cast<const vec4*>(&sh.q[4 * e]), q23 = *reinterpret_cast<const vec4*>(&sh.q[4 * e + 2]);
                        const cf ge = cadd(cadd(mk(q01.x, q01.y), mk(q01.z, q01.w)), cadd(mk(q23.x, q23.y), mk(q23.z, q23.w)));
                        sh.g[64 + e] = ge;
                        const cf xe = X[64 + e];
                        sh.dre[64 + e] = fma_(xe.x, ge.x, xe.y * ge.y);
                    }
                }
                const int wc = wc_of(l, r);
                if (wc >= 0 && l == wc * LPC) {                                  // the first lane of a channel's strip: the channel's prior error
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
            // ---- den, first level: sixteen lanes sum CNP / 16 terms each
            ex.phase_wave([&](int l, Rg&) {
                if (l < 16) {
                    float a = 0.0f;
#pragma unroll
                    for (int m = 0; m < CNP / 16; ++m) a += sh.dre[l * (CNP / 16) + m];
                    sh.p16[l] = a;
                }
            });
            // ---- gain (awpe.py:163-180): var, den; g is rescaled in place to h = g sqrt(1 / (den lambda)), so that the downdate below is
            // P / lambda - h_i conj(h_j): four packed instructions per element instead of five
            ex.phase_wave([&](int l, Rg& r) {
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
                sh.g[l] = cscale(sh.g[l], hs);
                if constexpr (XR > 0) { if ((l & 3) == 0) sh.g[64 + (l >> 2)] = cscale(sh.g[64 + (l >> 2)], hs); }
                if (l == 0) sh.ks = hs > 0.0f ? dinv / hs : 0.0f;              // kn = g / den = h (dinv / hs)
            });
            // ---- P and W updates (awpe.py:181-189)
            ex.phase_wave([&](int l, Rg& r) {
                const cf hi = sh.g[l];
#pragma unroll
                for (int j = 0; j < CNP; j += 2) {
                    const vec4 h2 = *reinterpret_cast<const vec4*>(&sh.g[j]);
                    r.Pa[j] = herm_downdate_h(r.Pa[j], hi, mk(h2.x, h2.y), lam_inv);
                    r.Pa[j + 1] = herm_downdate_h(r.Pa[j + 1], hi, mk(h2.z, h2.w), lam_inv);
                }
                if constexpr (XR > 0) {
                    const int cq0 = (l & 3) * QW;
                    const cf he = sh.g[64 + (l >> 2)];
#pragma unroll
                    for (int j = 0; j < QW; ++j) r.Pb[j] = herm_downdate_h(r.Pb[j], he, sh.g[cq0 + j], lam_inv);
                }
                const int wc = wc_of(l, r);
                if (wc >= 0) {
                    const cf e = sh.err[wc];
                    const float ks = sh.ks;
                    const int wi0 = wi0_of(l, r);
#pragma unroll
                    for (int s = 0; s < SLP; ++s)
                        if (s < SL) r.W[s] = cadd(r.W[s], cmulc(cscale(sh.g[wi0 + s], ks), e));     // W_c += conj(err_c) kn  (awpe.py:188-189)
                }
            });
            cur = nxt;
        }

        // ---- epilogue: the small parts, then the upper triangle chunk by chunk through the tile
        if constexpr (!TAIL)
            ex.phase_wave([&](int l, Rg& r) {
                const int wc = wc_of(l, r), wi0 = wi0_of(l, r);
                if (wc >= 0) {
#pragma unroll
                    for (int s = 0; s < SLP; ++s)
                        if (s < SL && wi0 + s < CN) st[Lb.w0 + wc * CN + wi0 + s] = r.W[s];
                }
                for (int i = l; i < CN; i += NT) st[Lb.x0 + i] = sh.X[cur][i];
                if (l == 0) stf[Lb.var_f] = r.var;
            });
        auto store_chunk = [&](auto hc) {
            constexpr int H = decltype(hc)::value, c0 = D::col0(H), c1r = D::col0(H + 1), w0 = wpew_words(c0);
            if (c0 >= CN) return;
            const int c1 = c1r < CN ? c1r : CN;
            constexpr bool WITH_TAIL = TAIL && H == NCH - 1;
            const int w1 = WITH_TAIL ? Lb.x0 + CN + 2 : wpew_words(c1);
            ex.phase_wave([&](int l, Rg& r) {
                if constexpr (WITH_TAIL) {                              // W, the taps, (var, 0), the padding: into the tile behind the triangle
                    const int wc = wc_of(l, r), wi0 = wi0_of(l, r);
                    if (wc >= 0) {
#pragma unroll
                        for (int s = 0; s < SLP; ++s)
                            if (s < SL && wi0 + s < CN) sh.tile[Lb.w0 - w0 + wc * CN + wi0 + s] = r.W[s];
                    }
                    for (int i2 = l; i2 < CN; i2 += NT) sh.tile[Lb.x0 - w0 + i2] = sh.X[cur][i2];
                    for (int g2 = wpe_packed(CN) + l; g2 < Lb.w0; g2 += NT) sh.tile[g2 - w0] = mk(0.0f, 0.0f);   // the gap up to W's line
                    if (l < 2) sh.tile[Lb.x0 + CN - w0 + l] = mk(l == 0 ? r.var : 0.0f, 0.0f);
                }
                const int i = l;
#pragma unroll
                for (int q = c0; q < c1r; ++q)
                    if (q < c1) *(i <= q ? &sh.tile[wpew_words(q) - w0 + i] : &sh.spare[l]) = r.Pa[q];     // rows below the diagonal: a spare word of the lane's own
                if constexpr (XR > 0 && c1r > 64) {
                    const int e = 64 + (l >> 2), cq0 = (l & 3) * QW;
#pragma unroll
                    for (int j = 0; j < QW; ++j) {
                        const int cq = cq0 + j;
                        *((cq >= c0 && cq < c1 && e <= cq) ? &sh.tile[wpew_words(cq) - w0 + e] : &sh.spare[l]) = r.Pb[j];
                    }
                }
            });
            ex.phase_wave([&](int l, Rg&) { tile_out(l, w0, w1); });
        };
        store_chunk(std::integral_constant<int, 0>());
        if constexpr (NCH > 1) store_chunk(std::integral_constant<int, 1>());
        if constexpr (NCH > 2) store_chunk(std::integral_constant<int, 2>());
        if constexpr (NCH > 3) store_chunk(std::integral_constant<int, 3>());
    }
};

}  // namespace ds
