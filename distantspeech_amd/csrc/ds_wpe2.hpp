// ds_wpe2.hpp — the RLS-WPE frame program of ds_wpe.hpp with TWO rows of P per lane (round 5): C N / 2 lanes share a bin, lane l keeps rows
// l and l + C N / 2 of P, the matching columns of W and taps of the input buffer.  For the compile-time shapes with C N = 16 or 8 and
// C <= C N / 2 (BASELINE config 4's 8 x 2; 4 x 4; 4 x 2).
//
// Why: with 10 s per call the one-row-per-lane kernel is bound by the LDS and the vector issue slots at once (profiles/r05j/cfg4_T312_compute.json:
// SQ_LDS_IDX_ACTIVE 0.84 of the CU's busy cycles, VALU 1.00) — every lane of a bin reads the whole tap buffer, the whole g = P x and the 16
// partial products from LDS, and computes the bin's denominator on its own.  Two rows per lane halve those reads and the per-lane work (bin
// split, bounds, addresses, denominator) per row of P; the rows' own arithmetic is the statement sequence of WpeEngine, operand for operand,
// so the results are bit-identical to it (tests: this engine against the generic one, outputs and exported state).
// The prediction filters W live as ROWS here: lane c < C keeps W[c][0 .. C N) (the one-row kernel keeps a column per lane).  The filter output
// err_c = d_c - sum_i conj(W[c][i]) X_i is then lane-local — the same products added in the same order as the one-row kernel's sum over the
// lanes' partial products, without the partial-product array and the err hand-off in LDS (29 of the ~60 LDS instructions per lane and frame,
// and the frame's third phase) — and so is the update W[c][i] += conj(err_c) kn_i with the g_i the downdate reads anyway.
// State layout, parameters and the phase structure (wave-local phases, Exec policy) are ds_wpe.hpp's.
#pragma once
#include "ds_wpe.hpp"

namespace ds {

template <int CT, int NTAPS> struct Wpe2Dims {
    static constexpr int C = CT, N = NTAPS, CN = CT * NTAPS, LH = CN / 2;
    static_assert(CN == 16 || CN == 8, "two rows per lane: C N = 16 or 8");
    static_assert(CT <= LH && CT <= WPE_CMAX, "a lane per channel for the frame's current spectra");
    static constexpr int BPW = WPE_NT / LH;
    // rows of the partial-product array: C of them are used as such; as the tile of the packed triangle it must hold wpe_layout's tri_words
    static constexpr int TW = wpe_layout(CT, NTAPS).tri_words;
    static constexpr int PR = (TW + CN + 1) / (CN + 2) > C ? (TW + CN + 1) / (CN + 2) : C;
};

template <int CT, int NTAPS> struct Wpe2Shared {
    typedef Wpe2Dims<CT, NTAPS> D;
    // (rows of X, num, dre are one bin's and 128 / 128 / 64 B apart: the 4 bins of a 32-lane group read two rows per bank, SQ_LDS_BANK_CONFLICT is
    // 0.44 of SQ_LDS_IDX_ACTIVE.  Round 6 padded the rows to 144 / 144 / 80 B — conflict-free by the bank rule, 16-byte aligned: cfg4 with
    // 10 s per call 8.45 M against 8.80 M frames/s, 12 B of scratch in the 8 x 2 kernel.  The phases are wave-local and their LDS cycles hide
    // behind the arithmetic; the odd strides cost address arithmetic and a register.  profiles/r06a/wpe_pad_ab.txt)
    cf X[2][D::BPW][D::CN];               // input buffer, double-buffered across frames
    cf d[D::BPW][D::C];
    alignas(16) cf part[D::BPW][D::PR][D::CN + 2];  // the tile the packed triangle of P passes through on its way in and out (ds_wpe.hpp's layout of it)
    cf num[D::BPW][D::CN];                // g_i = (P X)_i
    float dre[D::BPW][D::CN];             // Re(conj(X_i) g_i)
};

template <int CT, int NTAPS> struct Wpe2Regs {
    typedef Wpe2Dims<CT, NTAPS> D;
    cf P[2][D::CN];
    cf W[D::CN];                          // row l of W (lanes l < C)
    cf xin[2], din, err;
    float var;
    long long io0, ring0;
};

template <int CT, int NTAPS> struct WpeEngine2 {
    typedef Wpe2Dims<CT, NTAPS> D;
    typedef Wpe2Shared<CT, NTAPS> Sh;
    typedef Wpe2Regs<CT, NTAPS> Rg;
    static constexpr int NT = WPE_NT, C = D::C, N = D::N, CN = D::CN, LH = D::LH, BPW = D::BPW;
    static_assert(D::PR * (CN + 2) >= D::TW, "the tile holds the packed triangle");
    static_assert(D::PR * (CN + 2) >= C * CN, "... and, after it, the C rows of W");

    template <class Exec> static DS_HD void run(Exec& ex, const WpeParams& p, int blk, Sh& sh) {
        const int SB = wpe_bin_floats(C, N);
        const long long nbins = (long long)p.B * p.K;
        const float lam = p.lam, lam_inv = 1.0f / p.lam;
        const int ring_pos = p.dev_ring_pos ? p.dev_ring_pos[0] : p.ring_pos;
        // lane -> (bin slot s, lane l of the bin; rows l and l + LH); g = global bin
        auto slot = [&](int tid, int& s, int& l, long long& g, bool& on) {
            s = tid / LH; l = tid - s * LH;
            g = (long long)blk * BPW + s;
            on = g < nbins;
        };
        auto bin_state = [&](long long g) {
            const long long b = g / p.K, k = g - b * p.K;
            return p.state + b * p.ustride + k * SB;
        };
        const long long fstride = (long long)p.K * C;
        auto io_at = [&](const Rg& r, int t) { return r.io0 + (long long)t * fstride; };
        auto ring_slot = [&](const Rg& r, int slot_) { return r.ring0 + (long long)slot_ * fstride; };
        auto delayed = [&](const Rg& r, int t, int c) {           // x_delayed[c] of frame t (one load from a selected address: ds_wpe.hpp)
            const float* src = p.xd;
            long long f = io_at(r, t);
            if (p.ring != nullptr) {
                const bool in_ring = t < p.ring_len;
                src = in_ring ? p.ring : p.d;
                f = in_ring ? ring_slot(r, (ring_pos + t) % p.ring_len) : io_at(r, t - p.ring_len);
            }
            return mk(src[2 * (f + c)], src[2 * (f + c) + 1]);
        };
        const WpeLayout Lb = wpe_layout(C, N);
        ex.phase_wave([&](int tid, Rg& r) {
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on) {
#pragma unroll
                for (int h = 0; h < 2; ++h) sh.X[0][s][l + h * LH] = mk(0.0f, 0.0f);
                return;
            }
            const float* stf = bin_state(g);
            const cf* st = reinterpret_cast<const cf*>(stf);
            cf* tri = &sh.part[s][0][0];
            if (Lb.lines) {
                for (int w = 2 * l; w < Lb.tri_words; w += 2 * LH) *reinterpret_cast<vec4*>(&tri[w]) = load_state(reinterpret_cast<const vec4*>(&st[w]));
#pragma unroll
                for (int h = 0; h < 2; ++h) sh.X[0][s][l + h * LH] = load_state(&st[Lb.x0 + l + h * LH]);
                r.err = mk(0.0f, 0.0f);
                r.var = 0.0f;                                     // (arrives with the tile: taken in the next phase)
            } else {
                if ((Lb.tri_words & 1) == 0) {
                    for (int w = 2 * l; w < Lb.tri_words; w += 2 * LH) *reinterpret_cast<vec4*>(&tri[w]) = *reinterpret_cast<const vec4*>(&st[w]);
                } else {
                    for (int w = l; w < Lb.tri_words; w += LH) tri[w] = st[w];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) sh.X[0][s][l + h * LH] = st[Lb.x0 + l + h * LH];
                r.err = mk(0.0f, 0.0f);
                r.var = stf[Lb.var_f];
            }
            const long long b = g / p.K, k = g - b * p.K;
            r.io0 = ((b * p.T + 0) * p.K + k) * C;
            r.ring0 = p.ring != nullptr ? ((b * p.ring_len + 0) * p.K + k) * C : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) r.xin[h] = delayed(r, 0, (l + h * LH) / N);
            if (l < C) r.din = mk(p.d[2 * (r.io0 + l)], p.d[2 * (r.io0 + l) + 1]);
        });
        ex.phase_wave([&](int tid, Rg& r) {                            // rows l, l + LH of P: above the diagonal as stored, below it the conjugate of the column
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on) return;
            const cf* tri = &sh.part[s][0][0];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = l + h * LH;
#pragma unroll
                for (int q = 0; q < CN; ++q) r.P[h][q] = q >= i ? tri[q * (q + 1) / 2 + i] : cconj(tri[i * (i + 1) / 2 + q]);
            }
            if (CN == WPE_CNMAX) r.var = tri[wpe_packed(WPE_CNMAX)].x;     // the line layout keeps var behind the triangle (wpe_layout)
        });
        // W, C rows of C N words back to back in the block: through the same tile (coalesced 16-byte pieces on the HBM side), then lane l < C takes row l
        constexpr bool WV4 = (wpe_layout(C, N).w0 & 1) == 0;
        ex.phase_wave([&](int tid, Rg& r) {
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on) return;
            const cf* st = reinterpret_cast<const cf*>(bin_state(g));
            cf* tri = &sh.part[s][0][0];
            if (WV4) {
                for (int w = 2 * l; w < C * CN; w += 2 * LH) *reinterpret_cast<vec4*>(&tri[w]) = load_state(reinterpret_cast<const vec4*>(&st[Lb.w0 + w]));
            } else {
                for (int w = l; w < C * CN; w += LH) tri[w] = st[Lb.w0 + w];
            }
        });
        ex.phase_wave([&](int tid, Rg& r) {
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on) return;
            const cf* tri = &sh.part[s][0][0];
#pragma unroll
            for (int q = 0; q < CN; ++q) r.W[q] = l < C ? tri[l * CN + q] : mk(0.0f, 0.0f);
        });
        int cur = 0;
        for (int t = 0; t < p.T; ++t) {
            const int nxt = cur ^ 1;
            // ---- buffer_input (:80-102): per channel shift along the taps, newest delayed frame at tap 0
            ex.phase_wave([&](int tid, Rg& r) {
                int s, l; long long g; bool on;
                slot(tid, s, l, g, on);
                if (!on) return;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = l + h * LH;
                    const bool tap0 = i == (i / N) * N;
                    const cf prev = sh.X[cur][s][i > 0 ? i - 1 : 0], xin = r.xin[h];
                    sh.X[nxt][s][i] = mk(tap0 ? xin.x : prev.x, tap0 ? xin.y : prev.y);
                }
                if (l < C) {
                    sh.d[s][l] = r.din;
                    if (p.ring != nullptr && t >= p.T - p.ring_len) {             // this frame is one of the last ring_len: keep it
                        const long long f = ring_slot(r, (ring_pos + t) % p.ring_len);
                        p.ring[2 * (f + l)] = r.din.x; p.ring[2 * (f + l) + 1] = r.din.y;
                    }
                }
            });
            // ---- per-row products g_i = (P X)_i and Re(conj(X_i) g_i); err_l = d_l - sum_i conj(W[l][i]) X_i (:158-161: the products one by one,
            // added in tap order — what the one-row kernel's lanes hand to their channel's lane through LDS).  The tap buffer is read once for all of it
            ex.phase_wave([&](int tid, Rg& r) {
                int s, l; long long g; bool on;
                slot(tid, s, l, g, on);
                if (!on) return;
                cf a0[2] = {mk(0.0f, 0.0f), mk(0.0f, 0.0f)}, a1[2] = {mk(0.0f, 0.0f), mk(0.0f, 0.0f)};
                cf o = mk(0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < CN; j += 2) {
                    const cf x0 = sh.X[nxt][s][j], x1 = sh.X[nxt][s][j + 1];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        a0[h] = cfma(a0[h], r.P[h][j], x0);
                        a1[h] = cfma(a1[h], r.P[h][j + 1], x1);
                    }
                    o = cadd(o, cmulc(x0, r.W[j]));
                    o = cadd(o, cmulc(x1, r.W[j + 1]));
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = l + h * LH;
                    const cf Xi = sh.X[nxt][s][i];
                    const cf a = cadd(a0[h], a1[h]);
                    sh.num[s][i] = a;
                    sh.dre[s][i] = fma_(Xi.x, a.x, Xi.y * a.y);
                }
                if (l < C) {
                    const cf e = csub(r.din, o);
                    r.err = e;
                    const long long f = io_at(r, t);
                    p.err[2 * (f + l)] = e.x; p.err[2 * (f + l) + 1] = e.y;
                    if (p.err0 != nullptr && l == 0) { const long long f0 = f / C; p.err0[2 * f0] = e.x; p.err0[2 * f0 + 1] = e.y; }   // [B][T][K]
                }
                if (t + 1 < p.T) {                                 // next frame's inputs: in flight behind this frame's arithmetic
                    const long long f1 = io_at(r, t + 1);
#pragma unroll
                    for (int h = 0; h < 2; ++h) r.xin[h] = delayed(r, t + 1, (l + h * LH) / N);
                    if (l < C) r.din = mk(p.d[2 * (f1 + l)], p.d[2 * (f1 + l) + 1]);
                }
            });
            // ---- gain, P and W updates (g read once for both rows)
            ex.phase_wave([&](int tid, Rg& r) {
                int s, l; long long g; bool on;
                slot(tid, s, l, g, on);
                if (!on) return;
                float dpow = 0.0f;
                for (int c = 0; c < C; ++c) dpow += cabs2(sh.d[s][c]);
                r.var = fma_(0.98f, r.var, (float)(1.0 - 0.98) * (dpow / (float)C));       // :163-165
                float den = lam * r.var;
                for (int q = 0; q < CN; ++q) den += sh.dre[s][q];                          // :174-180, real part (ds_wpe.hpp)
                const float dinv = den == 0.0f ? 0.0f : 1.0f / den;                        // (digital silence: ds_wpe.hpp)
                const float dls = dinv * lam_inv;
                const cf gi[2] = {sh.num[s][l], sh.num[s][l + LH]};                         // (re-read: four registers less across the phase boundary)
#pragma unroll
                for (int j = 0; j < CN; ++j) {
                    const cf gj = sh.num[s][j];
#pragma unroll
                    for (int h = 0; h < 2; ++h) r.P[h][j] = herm_downdate(r.P[h][j], gi[h], gj, lam_inv, dls);     // :183-185 (ds_wpe.hpp)
                    r.W[j] = cadd(r.W[j], cmulc(cscale(gj, dinv), r.err));                                            // W_l += conj(err_l) kn  :188-189
                }
            });
            cur = nxt;
        }
        ex.phase_wave([&](int tid, Rg& r) {
            DS_PIN(tid);                                                   // (once per call: the bin's address and the lane's place in it formed again here, not held across the frame loop)                            // W back: rows into the tile ...
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on || l >= C) return;
            cf* tri = &sh.part[s][0][0];
#pragma unroll
            for (int q = 0; q < CN; ++q) tri[l * CN + q] = r.W[q];
        });
        ex.phase_wave([&](int tid, Rg& r) {                            // ... and out of it in coalesced pieces
            DS_PIN(tid);
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on) return;
            cf* st = reinterpret_cast<cf*>(bin_state(g));
            const cf* tri = &sh.part[s][0][0];
            if (WV4) {
                if (Lb.lines) { for (int w = 2 * l; w < C * CN; w += 2 * LH) store_state(reinterpret_cast<vec4*>(&st[Lb.w0 + w]), *reinterpret_cast<const vec4*>(&tri[w])); }
                else { for (int w = 2 * l; w < C * CN; w += 2 * LH) *reinterpret_cast<vec4*>(&st[Lb.w0 + w]) = *reinterpret_cast<const vec4*>(&tri[w]); }
            } else {
                for (int w = l; w < C * CN; w += LH) st[Lb.w0 + w] = tri[w];
            }
        });
        ex.phase_wave([&](int tid, Rg& r) {                            // the upper triangle back through the tile
            DS_PIN(tid);
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on) return;
            cf* tri = &sh.part[s][0][0];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = l + h * LH;
#pragma unroll
                for (int q = 0; q < CN; ++q)
                    if (q >= i) tri[q * (q + 1) / 2 + i] = r.P[h][q];
            }
            if (CN == WPE_CNMAX) tri[wpe_packed(WPE_CNMAX) + l] = mk(l == 0 ? r.var : 0.0f, 0.0f);       // (var, 0) and the line's padding (LH = 8 words)
        });
        ex.phase_wave([&](int tid, Rg& r) {
            DS_PIN(tid);
            int s, l; long long g; bool on;
            slot(tid, s, l, g, on);
            if (!on) return;
            float* stf = bin_state(g);
            cf* st = reinterpret_cast<cf*>(stf);
            const cf* tri = &sh.part[s][0][0];
            if (Lb.lines) {                                       // (var went into the tile with the rows, one phase back)
                for (int w = 2 * l; w < Lb.tri_words; w += 2 * LH) store_state(reinterpret_cast<vec4*>(&st[w]), *reinterpret_cast<const vec4*>(&tri[w]));
#pragma unroll
                for (int h = 0; h < 2; ++h) store_state(&st[Lb.x0 + l + h * LH], sh.X[cur][s][l + h * LH]);
            } else {
                if ((Lb.tri_words & 1) == 0) {
                    for (int w = 2 * l; w < Lb.tri_words; w += 2 * LH) *reinterpret_cast<vec4*>(&st[w]) = *reinterpret_cast<const vec4*>(&tri[w]);
                } else {
                    for (int w = l; w < Lb.tri_words; w += LH) st[w] = tri[w];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) st[Lb.x0 + l + h * LH] = sh.X[cur][s][l + h * LH];
                if (l == 0) stf[Lb.var_f] = r.var;
            }
        });
    }
};

}  // namespace ds
