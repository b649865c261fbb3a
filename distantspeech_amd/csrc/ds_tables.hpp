// ds_tables.hpp — host-side constant tables (twiddles, sqrt-Hann window, ISTFT scale).
#pragma once
#include <cmath>
#include <vector>

#include "ds_core.hpp"

namespace ds {

// twN[i] = exp(-2 pi j i / N), i = 0..N/2 ; win = sqrt(periodic Hann) (transform/transform.py:418-419);
// out_scale = hop / sum(win^2) (transform.py:428,479).  Computed in double, rounded once.
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

}  // namespace ds
