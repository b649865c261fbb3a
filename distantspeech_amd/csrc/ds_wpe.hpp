// ds_wpe.hpp — RLS-based online WPE, the frequency-domain core of Wpe.update (dereverberation/awpe.py:129-192) on the STFT grid.
//   x_delayed complex [B][T][K][C] (the frame from `delay` hops ago), d complex [B][T][K][C] (current frame)
//   -> err complex [B][T][K][C] (dereverberated frame, all channels)
// One (utterance, bin) is a CN x CN complex RLS (CN = C * N taps-by-channels, <= 16).  CN lanes share a bin: lane i keeps row i
// of P, column i of W and tap i of the input buffer in registers for all T frames of the call; the three reductions of a frame
// (W^H X, X^H P X, X^H P) go through LDS in a fixed order.  A workgroup of 128 lanes carries 128 / LPB bins (LPB = CN rounded up
// to 4, 8 or 16).  State per bin is one contiguous block laid out slot-major, lane-minor, so that the CN lanes of a bin read
// consecutive 8-byte words: every state access is a fully used 128-byte segment (CN = 16), once in and once out per call.
// Written against the Exec policy (tests/emul runs it serially on the CPU).
#pragma once
#include "ds_core.hpp"

namespace ds {

constexpr int WPE_CNMAX = 16, WPE_CMAX = 8, WPE_NT = 128;

// per-bin block: NQ = CN + C + 1 slots of CN complex, element (slot q, lane i) at 2 * (q * CN + i):
//   q < CN: P[i][q]      CN <= q < CN + C: W[q - CN][i]      q = CN + C: input_buffer tap i      then var (1 float), padded to 16 B
DS_HD constexpr int wpe_slots(int C, int N) { return C * N + C + 1; }
DS_HD constexpr int wpe_bin_floats(int C, int N) { return (wpe_slots(C, N) * C * N * 2 + 1 + 3) & ~3; }
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
};

template <int LPB> struct WpeShared {
    static constexpr int BPW = WPE_NT / LPB, CM = LPB < WPE_CMAX ? LPB : WPE_CMAX;
    cf X[2][BPW][LPB];               // input buffer, double-buffered across frames (tap shift reads the neighbour lane)
    cf d[BPW][CM];
    cf part[BPW][CM][LPB + 1];       // conj(W[c][i]) X_i
    cf dpart[BPW][LPB];              // conj(X_i) (P X)_i
    cf xhp[BPW][LPB][LPB + 1];       // [j][i] = conj(X_i) P[i][j]
    cf xh[BPW][LPB];                 // (X^H P)_j
    cf err[BPW][CM];
};

template <int LPB> struct WpeRegs {
    static constexpr int CM = LPB < WPE_CMAX ? LPB : WPE_CMAX;
    cf P[LPB];
    cf W[CM];
    cf num, xin, din;
    float var;
};

template <int LPB> struct WpeEngine {
    static constexpr int NT = WPE_NT, BPW = WPE_NT / LPB, CM = LPB < WPE_CMAX ? LPB : WPE_CMAX;
    typedef WpeShared<LPB> Sh;
    typedef WpeRegs<LPB> Rg;

    template <class Exec> static DS_HD void run(Exec& ex, const WpeParams& p, int blk, Sh& sh) {
        const int C = p.C, N = p.N, CN = C * N;
        const int SB = wpe_bin_floats(C, N);
        const long long nbins = (long long)p.B * p.K;
        const float lam = p.lam, lam_inv = 1.0f / p.lam;
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
        auto delayed = [&](long long g, int t, int c) {           // x_delayed[c] of frame t
            if (p.ring == nullptr) { const long long f = io_base(g, t); return mk(p.xd[2 * (f + c)], p.xd[2 * (f + c) + 1]); }
            if (t < p.ring_len) { const long long f = ring_at(g, (p.ring_pos + t) % p.ring_len); return mk(p.ring[2 * (f + c)], p.ring[2 * (f + c) + 1]); }
            const long long f = io_base(g, t - p.ring_len);
            return mk(p.d[2 * (f + c)], p.d[2 * (f + c) + 1]);
        };
        ex.phase([&](int tid, Rg& r) {
            int s, i; long long g; bool on;
            slot(tid, s, i, g, on);
            if (!on) { if (i < LPB && s < BPW) sh.X[0][s][i] = mk(0.0f, 0.0f); return; }
            const float* stf = bin_state(g);
            const cf* st = reinterpret_cast<const cf*>(stf);
#pragma unroll
            for (int q = 0; q < LPB; ++q)
                if (q < CN) r.P[q] = st[q * CN + i];
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < C) r.W[c] = st[(CN + c) * CN + i];
            sh.X[0][s][i] = st[(CN + C) * CN + i];
            r.var = stf[2 * wpe_slots(C, N) * CN];
            const long long f0 = io_base(g, 0);
            const int c = i / N;
            if (i == c * N) r.xin = delayed(g, 0, c);
            if (i < C) r.din = mk(p.d[2 * (f0 + i)], p.d[2 * (f0 + i) + 1]);
        });
        int cur = 0;
        for (int t = 0; t < p.T; ++t) {
            const int nxt = cur ^ 1;
            // ---- buffer_input (:80-102): per channel shift along the taps, newest delayed frame at tap 0
            ex.phase([&](int tid, Rg& r) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on) return;
                const int c = i / N;
                sh.X[nxt][s][i] = (i == c * N) ? r.xin : sh.X[cur][s][i - 1];
                if (i < C) {
                    sh.d[s][i] = r.din;
                    if (p.ring != nullptr && t >= p.T - p.ring_len) {             // this frame is one of the last ring_len: keep it
                        const long long f = ring_at(g, (p.ring_pos + t) % p.ring_len);
                        p.ring[2 * (f + i)] = r.din.x; p.ring[2 * (f + i) + 1] = r.din.y;
                    }
                }
            });
            // ---- per-lane products: (P X)_i, conj(W[c][i]) X_i, conj(X_i) (P X)_i, conj(X_i) P[i][j]
            ex.phase([&](int tid, Rg& r) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on) return;
                const cf Xi = sh.X[nxt][s][i];
                cf a = mk(0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < LPB; ++j)
                    if (j < CN) {
                        a = cfma(a, r.P[j], sh.X[nxt][s][j]);
                        sh.xhp[s][j][i] = cmulc(r.P[j], Xi);
                    }
                r.num = a;
                sh.dpart[s][i] = cmulc(a, Xi);
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) sh.part[s][c][i] = cmulc(Xi, r.W[c]);
                if (t + 1 < p.T) {                                 // next frame's inputs: in flight behind this frame's arithmetic
                    const long long f1 = io_base(g, t + 1);
                    const int c = i / N;
                    if (i == c * N) r.xin = delayed(g, t + 1, c);
                    if (i < C) r.din = mk(p.d[2 * (f1 + i)], p.d[2 * (f1 + i) + 1]);
                }
            });
            // ---- reductions in lane order: err_c = d_c - sum_i ..., (X^H P)_j = sum_i ...
            ex.phase([&](int tid, Rg&) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on) return;
                cf acc = mk(0.0f, 0.0f);
                for (int l = 0; l < CN; ++l) acc = cadd(acc, sh.xhp[s][i][l]);
                sh.xh[s][i] = acc;
                if (i < C) {                                       // err = d - W^H X  (:158-161)
                    cf o = mk(0.0f, 0.0f);
                    for (int l = 0; l < CN; ++l) o = cadd(o, sh.part[s][i][l]);
                    const cf e = csub(sh.d[s][i], o);
                    sh.err[s][i] = e;
                    const long long f = io_base(g, t);
                    p.err[2 * (f + i)] = e.x; p.err[2 * (f + i) + 1] = e.y;
                }
            });
            // ---- gain, P and W updates
            ex.phase([&](int tid, Rg& r) {
                int s, i; long long g; bool on;
                slot(tid, s, i, g, on);
                if (!on) return;
                float dpow = 0.0f;
                for (int c = 0; c < C; ++c) dpow += cabs2(sh.d[s][c]);
                r.var = fma_(0.98f, r.var, (float)(1.0 - 0.98) * (dpow / (float)C));       // :163-165
                cf den = mk(lam * r.var, 0.0f);
                for (int l = 0; l < CN; ++l) den = cadd(den, sh.dpart[s][l]);              // :174-180
                // digital silence from the first frame on (var = 0, X = 0) makes the reference's gain 0 / 0 and its state NaN for good; the
                // gain is 0 there instead — the only departure from awpe.py:174-180, and only where the reference has no finite value
                const cf kn = (den.x == 0.0f && den.y == 0.0f) ? mk(0.0f, 0.0f) : cdiv(r.num, den);
#pragma unroll
                for (int j = 0; j < LPB; ++j)
                    if (j < CN) r.P[j] = cscale(cfnma(r.P[j], kn, sh.xh[s][j]), lam_inv);    // P = (P - kn (X^H P)) / lambda  :183-185
#pragma unroll
                for (int c = 0; c < CM; ++c)
                    if (c < C) r.W[c] = cadd(r.W[c], cmulc(kn, sh.err[s][c]));             // W_c += conj(err_c) kn  :188-189
            });
            cur = nxt;
        }
        ex.phase([&](int tid, Rg& r) {
            int s, i; long long g; bool on;
            slot(tid, s, i, g, on);
            if (!on) return;
            float* stf = bin_state(g);
            cf* st = reinterpret_cast<cf*>(stf);
#pragma unroll
            for (int q = 0; q < LPB; ++q)
                if (q < CN) st[q * CN + i] = r.P[q];
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < C) st[(CN + c) * CN + i] = r.W[c];
            st[(CN + C) * CN + i] = sh.X[cur][s][i];
            if (i == 0) stf[2 * wpe_slots(C, N) * CN] = r.var;
        });
    }
};

}  // namespace ds
