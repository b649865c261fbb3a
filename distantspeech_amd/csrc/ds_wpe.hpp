// ds_wpe.hpp — RLS-based online WPE, the frequency-domain core of Wpe.update (dereverberation/awpe.py:129-192) on the STFT grid.
//   x_delayed complex [B][T][K][C] (the frame from `delay` hops ago), d complex [B][T][K][C] (current frame)
//   -> err complex [B][T][K][C] (dereverberated frame, all channels)
// One (utterance, bin) is a CN x CN complex RLS (CN = C * N taps-by-channels, <= 16).  CN lanes share a bin: lane i keeps row i
// of P, column i of W and tap i of the input buffer in registers for all T frames of the call.  A workgroup of 128 lanes carries
// 128 / LPB bins (LPB = CN rounded up to 4, 8 or 16).
//
// P, the inverse correlation matrix, is Hermitian, and the recursion is written so that it stays Hermitian bit for bit: with
// g = P x the update is P <- (P - g g^H / den) / lambda, den = lambda var + Re(x^H g) — the reference's (P - k x^H P) / lambda with
// x^H P replaced by (P x)^H and the rounding-level imaginary part of den dropped (awpe.py:174-185; equal in exact arithmetic).  That
// buys three things: the third reduction of a frame (x^H P, a 16 x 16 exchange through LDS, more than half of the kernel's LDS)
// is gone; only the upper triangle of P is state (1.1 KB instead of 2 KB per bin in HBM, read and written once per call: the
// kernel is HBM-bound at one frame per call); and P cannot drift away from Hermitian over a long stream.
// State per bin is one contiguous block; every lane reads the words it needs straight from it (its row above the diagonal, and
// its column above the diagonal for the part of the row below it — no exchange), once in and once out per call.
// A bin's lanes never leave their wavefront (LPB divides 64) and every LDS hand-off is between the lanes of one bin, so no phase needs a
// workgroup barrier: they are wave-local phases (Exec::phase_wave), which is what the kernel lives on when a call carries many frames.
// Written against the Exec policy (tests/emul runs it serially on the CPU).
#pragma once
#include "ds_core.hpp"

namespace ds {

constexpr int WPE_CNMAX = 16, WPE_CMAX = 8, WPE_NT = 128;
constexpr int WPEW_CNMAX = 80;          // ... and up to this many taps-by-channels through the wavefront-per-bin program (ds_wpe_wide.hpp)

// per-bin block, complex words: P upper triangle by columns, P[i][q] (i <= q) at q (q + 1) / 2 + i; then W[c][i] at w0 + c CN + i;
// then input_buffer tap i at x0 + i; var (1 float) at float var_f.
//   C N < 16: everything back to back (w0 = NPK, x0 = NPK + C CN, var behind the taps), padded to 16 B.  C N > 16: see below.
//   C N == 16 — the 16-lane kernel's full shape: BASELINE config 4's 8 x 2, and 4 x 4 — every piece one instruction moves is whole
//   128-byte lines: [P: 136 words][(var, 0)][7 words of padding] = 144 words = 9 lines, then C rows of W (16 words = one line each), then
//   the taps (one line): 2304 B at 8 channels against 2244 B packed.  With the pieces on line boundaries the block's traffic can be
//   non-temporal (on the packed block that policy COST 4 - 11 %: a line touched by two instructions was fetched twice): the same
//   pieces moved with no arithmetic run 19 % faster (scratch/micro/block_rw.hip `narrow`, profiles/r04a/block_rw_narrow.txt).
DS_HD constexpr int wpe_packed(int CN) { return CN * (CN + 1) / 2; }
struct WpeLayout { int w0, x0, var_f, floats, tri_words; bool lines; };
DS_HD constexpr WpeLayout wpe_layout(int C, int N) {
    const int CN = C * N, NPK = wpe_packed(CN);
    if (CN == 16) return WpeLayout{144, 144 + C * 16, 2 * 136, 2 * (144 + C * 16 + 16), 144, true};
    // the wide kernel's blocks (ds_wpe_wide.hpp): the block and its W section start on 128-byte lines (29 184 B at 4 x 20 against 29 124 B
    // packed) — the 1 KB pieces of the tile traffic are then whole lines in every block, not in every other one: +5 % for the access
    // pattern alone (profiles/r04a/block_rw.txt, `blk 29184`)
    if (CN > 16) { const int w0 = (NPK + 15) & ~15, x0 = w0 + C * CN; return WpeLayout{w0, x0, 2 * (x0 + CN), (2 * (x0 + CN) + 1 + 31) & ~31, NPK, false}; }
    return WpeLayout{NPK, NPK + C * CN, 2 * (NPK + C * CN + CN), (2 * (NPK + C * CN + CN) + 1 + 3) & ~3, NPK, false};
}
DS_HD constexpr int wpe_bin_floats(int C, int N) { return wpe_layout(C, N).floats; }
// the bytes of a block that carry state (what a launch must move once in and once out; the line layout's padding is not among them)
DS_HD constexpr int wpe_bin_floats_packed(int C, int N) { return 2 * (wpe_packed(C * N) + C * C * N + C * N) + 1; }
DS_HD constexpr int wpe_lanes_per_bin(int CN) { return CN <= 4 ? 4 : CN <= 8 ? 8 : 16; }

struct WpeParams {
    int B, K, T, C, N;
    const float* xd;
    const float* d;
    float* err;
    float* state;        // utterance b, bin k at state + b * ustride + k * wpe_bin_floats
    long long ustride;   // floats between utterances (>= K * wpe_bin_floats)
    float lam;           // forgetting factor (awpe.py:33)
    // optional frame delay line (the DS_ALGO_WPE_MVDR chain): when ring != null, xd is not read; frame t takes its delayed
    // input from ring slot (ring_pos + t) % ring_len for t < ring_len and from d[t - ring_len] after that, and the last
    // ring_len frames of d are left in the ring (the caller advances ring_pos by T)
    float* ring;         // complex [B][ring_len][K][C]
    int ring_pos, ring_len;
    const int* dev_ring_pos;   // optional device-resident ring position (graph replay); overrides ring_pos
    float* err0;         // optional: channel 0 of err alone, complex [B][T][K] (the input of a single-channel synthesis; ds_wpe_wide.hpp)
};

template <int LPB> struct WpeShared {
    static constexpr int BPW = WPE_NT / LPB, CM = LPB < WPE_CMAX ? LPB : WPE_CMAX;
    cf X[2][BPW][LPB];               // input buffer, double-buffered across frames (tap shift reads the neighbour lane)
    cf d[BPW][CM];
    alignas(16) cf part[BPW][CM][LPB + 2];   // conj(W[c][i]) X_i; doubles as the tile the packed triangle of P passes through (144 words at LPB = 16)
    cf num[BPW][LPB];                // g_i = (P X)_i
    float dre[BPW][LPB];             // Re(conj(X_i) g_i)
    cf err[BPW][CM];
};

template <int LPB> struct WpeRegs {
    static constexpr int CM = LPB < WPE_CMAX ? LPB : WPE_CMAX;
    cf P[LPB];
    cf W[CM];
    cf num, xin, din;
    float var;
    long long io0, ring0;          // channel 0 of frame 0 / of ring slot 0 of this lane's bin in the [B][T][K][C] / [B][ring_len][K][C] arrays
};

// CT, NTAPS > 0: the channel and tap counts as compile-time constants (the shapes the launcher knows: launch_wpe) — the `c < C`, `j < CN`
// guards of the generic program fold away with their scalar branches (29 % of the generic kernel's instruction stream at 8 x 2), the loops
// unroll to the live entries only.  Same statements, same order: bit-identical to the generic instantiation (test_wpe_*).
template <int LPB, int CT = 0, int NTAPS = 0> struct WpeEngine {
    static constexpr int NT = WPE_NT, BPW = WPE_NT / LPB, CM = LPB < WPE_CMAX ? LPB : WPE_CMAX;
    typedef WpeShared<LPB> Sh;
    typedef WpeRegs<LPB> Rg;
    static_assert(CT * NTAPS <= LPB && CT <= CM, "shape");

    template <class Exec> static DS_HD void run(Exec& ex, const WpeParams& p, int blk, Sh& sh) {
        const int C = CT > 0 ? CT : p.C, N = NTAPS > 0 ? NTAPS : p.N, CN = C * N;
        const int SB = wpe_bin_floats(C, N);
        const long long nbins = (long long)p.B * p.K;
        const float lam = p.lam, lam_inv = 1.0f / p.lam;
        const int ring_pos = p.dev_ring_pos ? p.dev_ring_pos[0] : p.ring_pos;
        // lane -> (bin slot s, lane i of the bin); g = global bin
        auto slot = [&](int tid, int& s, int& i, long long& g, bool& on) {
            s = tid / LPB; i = tid - s * LPB;
            g = (long long)blk * BPW + s;
            on = i < CN && g < nbins;
        };
        auto bin_state = [&](long long g) {
            const long long b = g / p.K, k = g - b * p.K;
            return p.state + b * p.ustride + k * SB;
        };
        auto io_base = [&](long long g, int t) {                  // index of channel 0 of frame t of bin g in the [B][T][K][C] arrays
            const long long b = g / p.K, k = g - b * p.K;
            return ((b * p.T + t) * p.K + k) * C;
        };
        auto ring_at = [&](long long g, int slot_) {              // channel 0 of ring slot `slot_` of bin g
            const long long b = g / p.K, k = g - b * p.K;
            return ((b * p.ring_len + slot_) * p.K + k) * C;
        };
        // the split of a bin index into (utterance, bin) is a 64-bit division by a run-time K: done once per lane (first phase), the frame
        // loop then steps through the arrays from the lane's frame-0 / slot-0 offsets (r.io0, r.ring0) by the uniform frame stride
        const long long fstride = (long long)p.K * C;
        auto io_at = [&](const Rg& r, int t) { return r.io0 + (long long)t * fstride; };
        auto ring_slot = [&](const Rg& r, int slot_) { return r.ring0 + (long long)slot_ * fstride; };
        // (one load from a selected address, not a load per source: three returns made the compiler keep the value in scratch and wait
        // for the load on the spot — the prefetch of the next frame's input was a stall)
        auto delayed = [&](const Rg& r, int t, int c) {           // x_delayed[c] of frame t
            const float* src = p.xd;
            long long f = io_at(r, t);
            if (p.ring != nullptr) {
                const bool in_ring = t < p.ring_len;
                src = in_ring ? p.ring : p.d;
                f = in_ring ? ring_slot(r, (ring_pos + t) % p.ring_len) : io_at(r, t - p.ring_len);
            }
            return mk(src[2 * (f + c)], src[2 * (f + c) + 1]);
        };
        ex.phase_wave([&](int tid, Rg& r) {
            int s, i; long long g; bool on;
            slot(tid, s, i, g, on);
            if (!on) { if (i < LPB && s < BPW) sh.X[0][s][i] = mk(0.0f, 0.0f); return; }
            const float* stf = bin_state(g);
            const cf* st = reinterpret_cast<const cf*>(stf);
            const WpeLayout Lb = wpe_layout(C, N);
            cf* tri = &sh.part[s][0][0];                          // the packed triangle passes through LDS (the `part` tile, idle here):
            // consecutive lanes, consecutive 16-byte pieces (an even word count; a block starts on a 16-byte boundary)
            if (Lb.lines) {                                       // whole lines per instruction, streamed (see wpe_layout)
                for (int w = 2 * i; w < Lb.tri_words; w += 2 * CN) *reinterpret_cast<vec4*>(&tri[w]) = load_state(reinterpret_cast<const vec4*>(&st[w]));
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) r.W[c] = load_state(&st[Lb.w0 + c * CN + i]);
                sh.X[0][s][i] = load_state(&st[Lb.x0 + i]);
                r.var = 0.0f;                                     // (arrives with the tile: taken in the next phase)
            } else {
                if ((Lb.tri_words & 1) == 0) {
                    for (int w = 2 * i; w < Lb.tri_words; w += 2 * CN) *reinterpret_cast<vec4*>(&tri[w]) = *reinterpret_cast<const vec4*>(&st[w]);
                } else {
                    for (int w = i; w < Lb.tri_words; w += CN) tri[w] = st[w];
                }
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) r.W[c] = st[Lb.w0 + c * CN + i];
                sh.X[0][s][i] = st[Lb.x0 + i];
                r.var = stf[Lb.var_f];
            }
            r.io0 = io_base(g, 0);
            r.ring0 = p.ring != nullptr ? ring_at(g, 0) : 0;
            const long long f0 = r.io0;
            r.xin = delayed(r, 0, i / N);          // every lane of a channel (only tap 0 uses it: the others would need a guarded assignment,
                                                   // which the compiler turns into a scratch slot and a wait on the load)
            if (i < C) r.din = mk(p.d[2 * (f0 + i)], p.d[2 * (f0 + i) + 1]);
        });
        ex.phase_wave([&](int tid, Rg& r) {                            // row i of P: above the diagonal as stored, below it the conjugate of column i
            int s, i; long long g; bool on;
            slot(tid, s, i, g, on);
            if (!on) return;
            const cf* tri = &sh.part[s][0][0];
#pragma unroll
            for (int q = 0; q < LPB; ++q)
                if (q < CN) r.P[q] = q >= i ? tri[q * (q + 1) / 2 + i] : cconj(tri[i * (i + 1) / 2 + q]);
            if (CN == WPE_CNMAX) r.var = tri[wpe_packed(WPE_CNMAX)].x;     // the line layout keeps var behind the triangle (wpe_layout)
        });
        int cur = 0;
        for (int t = 0; t < p.T; ++t) {
            const int nxt = cur ^ 1;
            // ---- buffer_input (:80-102): per channel shift along the taps, newest delayed frame at tap 0
            ex.phase_wave([&](int tid, Rg& r) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on) return;
                // (both candidates as values, then a select: `cond ? r.xin : sh.X[..]` on the structs became a select of ADDRESSES — a flat load
                // from either scratch or LDS, with r.xin parked in scratch and the prefetch waited for on the spot)
                const bool tap0 = i == (i / N) * N;
                const cf prev = sh.X[cur][s][i > 0 ? i - 1 : 0], xin = r.xin;
                sh.X[nxt][s][i] = mk(tap0 ? xin.x : prev.x, tap0 ? xin.y : prev.y);
                if (i < C) {
                    sh.d[s][i] = r.din;
                    if (p.ring != nullptr && t >= p.T - p.ring_len) {             // this frame is one of the last ring_len: keep it
                        const long long f = ring_slot(r, (ring_pos + t) % p.ring_len);
                        p.ring[2 * (f + i)] = r.din.x; p.ring[2 * (f + i) + 1] = r.din.y;
                    }
                }
            });
            // ---- per-lane products: g_i = (P X)_i, conj(W[c][i]) X_i, Re(conj(X_i) g_i)
            ex.phase_wave([&](int tid, Rg& r) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on) return;
                const cf Xi = sh.X[nxt][s][i];
                // two partial sums (even and odd taps) added at the end: half the dependent chain of multiply-adds per lane
                cf a0 = mk(0.0f, 0.0f), a1 = mk(0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < LPB; j += 2) {
                    if (j < CN) a0 = cfma(a0, r.P[j], sh.X[nxt][s][j]);
                    if (j + 1 < CN) a1 = cfma(a1, r.P[j + 1], sh.X[nxt][s][j + 1]);
                }
                const cf a = cadd(a0, a1);
                r.num = a;
                sh.num[s][i] = a;
                sh.dre[s][i] = fma_(Xi.x, a.x, Xi.y * a.y);
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) sh.part[s][c][i] = cmulc(Xi, r.W[c]);
                if (t + 1 < p.T) {                                 // next frame's inputs: in flight behind this frame's arithmetic
                    const long long f1 = io_at(r, t + 1);
                    r.xin = delayed(r, t + 1, i / N);
                    if (i < C) r.din = mk(p.d[2 * (f1 + i)], p.d[2 * (f1 + i) + 1]);
                }
            });
            // ---- err_c = d_c - sum_i conj(W[c][i]) X_i in lane order  (:158-161)
            ex.phase_wave([&](int tid, Rg& r) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on || i >= C) return;
                cf o = mk(0.0f, 0.0f);
                for (int l = 0; l < CN; ++l) o = cadd(o, sh.part[s][i][l]);
                const cf e = csub(sh.d[s][i], o);
                sh.err[s][i] = e;
                const long long f = io_at(r, t);
                p.err[2 * (f + i)] = e.x; p.err[2 * (f + i) + 1] = e.y;
                if (p.err0 != nullptr && i == 0) { const long long f0 = f / C; p.err0[2 * f0] = e.x; p.err0[2 * f0 + 1] = e.y; }   // [B][T][K]
            });
            // ---- gain, P and W updates
            ex.phase_wave([&](int tid, Rg& r) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on) return;
                float dpow = 0.0f;
                for (int c = 0; c < C; ++c) dpow += cabs2(sh.d[s][c]);
                r.var = fma_(0.98f, r.var, (float)(1.0 - 0.98) * (dpow / (float)C));       // :163-165
                float den = lam * r.var;
                for (int l = 0; l < CN; ++l) den += sh.dre[s][l];                          // :174-180, real part (see the header)
                // digital silence from the first frame on (var = 0, X = 0) makes the reference's gain 0 / 0 and its state NaN for good; the
                // gain is 0 there instead — the only departure from awpe.py:174-180 where the reference has a finite value to depart from
                const float dinv = den == 0.0f ? 0.0f : 1.0f / den;
                const cf kn = cscale(r.num, dinv);
                const cf gi = r.num;
                const float dls = dinv * lam_inv;
#pragma unroll
                for (int j = 0; j < LPB; ++j)
                    if (j < CN) {
                        // P = (P - g g^H / den) / lambda (:183-185).  Every product is rounded on its own and the two of a sum are then added,
                        // so that element (j, i), which lane j computes, is the exact conjugate of this one: P stays Hermitian bit for bit
                        // P / lambda - g_i conj(g_j) (1 / (den lambda)): the fused multiply-add keeps the symmetry too (it commutes with
                        // negation).  On the diagonal g_i conj(g_i) has the imaginary part x y - x y = +0 exactly and P_ii's is 0 from the
                        // initial state on (a real multiple of the identity), so it stays +0 without a per-lane select
                        r.P[j] = herm_downdate(r.P[j], gi, sh.num[s][j], lam_inv, dls);
                    }
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) r.W[c] = cadd(r.W[c], cmulc(kn, sh.err[s][c]));             // W_c += conj(err_c) kn  :188-189
            });
            cur = nxt;
        }
        ex.phase_wave([&](int tid, Rg& r) {                            // the upper triangle back through the tile
            int s, i; long long g; bool on;
            slot(tid, s, i, g, on);
            if (!on) return;
            cf* tri = &sh.part[s][0][0];
#pragma unroll
            for (int q = 0; q < LPB; ++q)
                if (q < CN && q >= i) tri[q * (q + 1) / 2 + i] = r.P[q];
            if (CN == WPE_CNMAX && i < 8) tri[wpe_packed(WPE_CNMAX) + i] = mk(i == 0 ? r.var : 0.0f, 0.0f);   // (var, 0) and the line's padding
        });
        ex.phase_wave([&](int tid, Rg& r) {
            int s, i; long long g; bool on;
            slot(tid, s, i, g, on);
            if (!on) return;
            float* stf = bin_state(g);
            cf* st = reinterpret_cast<cf*>(stf);
            const WpeLayout Lb = wpe_layout(C, N);
            const cf* tri = &sh.part[s][0][0];
            if (Lb.lines) {                                       // (var went into the tile with the rows, one phase back)
                for (int w = 2 * i; w < Lb.tri_words; w += 2 * CN) store_state(reinterpret_cast<vec4*>(&st[w]), *reinterpret_cast<const vec4*>(&tri[w]));
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) store_state(&st[Lb.w0 + c * CN + i], r.W[c]);
                store_state(&st[Lb.x0 + i], sh.X[cur][s][i]);
            } else {
                if ((Lb.tri_words & 1) == 0) {
                    for (int w = 2 * i; w < Lb.tri_words; w += 2 * CN) *reinterpret_cast<vec4*>(&st[w]) = *reinterpret_cast<const vec4*>(&tri[w]);
                } else {
                    for (int w = i; w < Lb.tri_words; w += CN) st[w] = tri[w];
                }
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) st[Lb.w0 + c * CN + i] = r.W[c];
                st[Lb.x0 + i] = sh.X[cur][s][i];
                if (i == 0) stf[Lb.var_f] = r.var;
            }
        });
    }
};

}  // namespace ds
