// ds_tables.hpp — host-side constant tables (twiddles, sqrt-Hann window, ISTFT scale).
#pragma once
#include <cmath>
#include <vector>

#include "ds_core.hpp"

namespace ds {

// twN[i] = exp(-2 pi j i / N), i = 0..N/2 ; win = sqrt(periodic Hann) (transform/transform.py:418-419);
// out_scale = hop / sum(win^2) (transform.py:428,479).  Computed in double, rounded once.
// stw[Ns + k] = (w1, w2) for the Stockham stage with sub-transform size Ns (4, 16, 64, 256), k < Ns:
// w_r = exp(-2 pi j k r / (Ns R)), R = 4 except the closing radix-2 stage of the 128- and 512-point plans.
inline void make_stage_twiddles(int N, std::vector<vec4>& stw) {
    const int NC = N / 2;
    const double PI = 3.14159265358979323846;
    stw.assign(NC, vec4{1.0f, 0.0f, 1.0f, 0.0f});
    for (int Ns = 4; Ns < NC; Ns *= 4) {
        const int R = (Ns * 4 <= NC) ? 4 : 2;
        for (int k = 0; k < Ns && Ns + k < NC; ++k) {
            const double a1 = -2.0 * PI * (double)k / (double)(Ns * R);
            vec4 w;
            w.x = (float)std::cos(a1); w.y = (float)std::sin(a1);
            w.z = R == 4 ? (float)std::cos(2.0 * a1) : 1.0f;
            w.w = R == 4 ? (float)std::sin(2.0 * a1) : 0.0f;
            stw[Ns + k] = w;
        }
    }
}

inline void make_tables(int N, int hop, std::vector<cf>& tw, std::vector<float>& win, float& out_scale) {
    const int NC = N / 2;
    const double PI = 3.14159265358979323846;
    tw.resize(NC + 1);
    win.resize(N);
    for (int i = 0; i <= NC; ++i) {
        const double a = -2.0 * PI * (double)i / (double)N;
        tw[i].x = (float)std::cos(a);
        tw[i].y = (float)std::sin(a);
    }
    tw[0].x = 1.0f; tw[0].y = 0.0f;
    tw[NC].x = -1.0f; tw[NC].y = 0.0f;
    if (NC % 2 == 0) { tw[NC / 2].x = 0.0f; tw[NC / 2].y = -1.0f; }
    double W0 = 0.0;
    for (int n = 0; n < N; ++n) {
        const double w = std::sqrt(0.5 - 0.5 * std::cos(2.0 * PI * (double)n / (double)N));
        win[n] = (float)w;
        W0 += w * w;
    }
    out_scale = (float)((double)hop / W0);
}

// The Tables<NFFT> blob exactly as the kernel's LDS copy expects it: [tw: N/2+2 cf][stw: NSTW vec4][win: N float].
inline void make_table_blob(int N, int hop, std::vector<float>& blob, float& out_scale) {
    const int NC = N / 2, NSTW = NC == 512 ? 512 : 128;
    std::vector<cf> tw;
    std::vector<float> win;
    std::vector<vec4> stw;
    make_tables(N, hop, tw, win, out_scale);
    make_stage_twiddles(N, stw);
    blob.assign((size_t)(NC + 2) * 2 + (size_t)NSTW * 4 + N, 0.0f);
    float* q = blob.data();
    for (int i = 0; i <= NC; ++i) { q[2 * i] = tw[i].x; q[2 * i + 1] = tw[i].y; }
    q += (NC + 2) * 2;
    for (int i = 0; i < NSTW && i < (int)stw.size(); ++i) { q[4 * i] = stw[i].x; q[4 * i + 1] = stw[i].y; q[4 * i + 2] = stw[i].z; q[4 * i + 3] = stw[i].w; }
    q += NSTW * 4;
    for (int n = 0; n < N; ++n) q[n] = win[n];
}

}  // namespace ds
