// complex_helpers_check.hip — test infrastructure (built and run by tests/test_gpu_ops.py::test_packed_complex_helpers_equal_their_scalar_definitions
// on the GPU box): the packed two-instruction forms of the complex products in ds_core.hpp (cmul, cmulc, cfma, cfmac, cfnma, cfnmac, the quarter-turn adds of the butterfly, herm_downdate) against the
// scalar expressions that define their rounding (cmul_s, ...), bit for bit, on random operands and on the awkward ones (signed zeros,
// denormals, huge and tiny magnitudes, infinities).  Prints "ok <n>" or the first mismatches; exit status 0 / 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include "ds_core.hpp"

using ds::cf;
__global__ void k_check(const cf* a, const cf* b, const cf* c, cf* out_pk, cf* out_s, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cf x = a[i], y = b[i], z = c[i];
    cf* p = out_pk + 12 * (size_t)i;
    cf* s = out_s + 12 * (size_t)i;
    p[0] = ds::cmul(x, y);       s[0] = ds::cmul_s(x, y);
    p[1] = ds::cmulc(x, y);      s[1] = ds::cmulc_s(x, y);
    p[2] = ds::cfma(z, x, y);    s[2] = ds::cfma_s(z, x, y);
    p[3] = ds::cfmac(z, x, y);   s[3] = ds::cfmac_s(z, x, y);
    p[4] = ds::cfnma(z, x, y);   s[4] = ds::cfnma_s(z, x, y);
    p[5] = ds::cfnmac(z, x, y);  s[5] = ds::cfnmac_s(z, x, y);
    p[6] = ds::cadd_jd<+1>(x, y); s[6] = ds::mk(x.x - y.y, x.y + y.x);      // x + j y
    p[7] = ds::cadd_jd<-1>(x, y); s[7] = ds::mk(x.x + y.y, x.y - y.x);      // x - j y
    p[8] = ds::cadd_c(x, y);      s[8] = ds::cadd(x, ds::cconj(y));
    p[9] = ds::csub_c(x, y);      s[9] = ds::csub(x, ds::cconj(y));
    p[10] = ds::cdiv_2j(x);       s[10] = ds::mk(0.5f * x.y, -0.5f * x.x);
    p[11] = ds::herm_downdate(z, x, y, x.x, y.y);  s[11] = ds::herm_downdate_s(z, x, y, x.x, y.y);    // the RLS-WPE element update (ds_wpe.hpp)
}

static bool same(float u, float v) {
    if (u != u && v != v) return true;                       // NaN == NaN (payload / sign of a NaN is not part of the contract)
    uint32_t a, b; std::memcpy(&a, &u, 4); std::memcpy(&b, &v, 4);
    return a == b;
}

int main() {
    const int n = 1 << 20;
    std::vector<cf> a(n), b(n), c(n);
    std::mt19937 rng(12345);
    std::normal_distribution<float> g(0.0f, 1.0f);
    const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, 1e-42f, -3e-45f, 1.17549435e-38f, 3.4e38f, -3.4e38f, 1e-20f, 1e20f,
                             __builtin_inff(), -__builtin_inff(), 0.5f, 3.0f};
    const int ns = sizeof special / sizeof special[0];
    for (int i = 0; i < n; ++i) {
        auto pick = [&](int salt) -> float {
            if (i < 4096) return special[(i * 7 + salt * 13 + (i >> 4) * salt) % ns];          // the awkward values, in every position
            const float s = (i & 1023) == 0 ? 1e-19f : (i & 511) == 0 ? 1e19f : 1.0f;              // products that underflow / overflow
            return g(rng) * s;
        };
        a[i] = ds::mk(pick(1), pick(2)); b[i] = ds::mk(pick(3), pick(4)); c[i] = ds::mk(pick(5), pick(6));
    }
    cf *da, *db, *dc, *dp, *dsr;
    if (hipMalloc(&da, n * sizeof(cf)) != hipSuccess) { printf("no device memory / no device\n"); return 2; }
    hipMalloc(&db, n * sizeof(cf)); hipMalloc(&dc, n * sizeof(cf)); hipMalloc(&dp, 12 * (size_t)n * sizeof(cf)); hipMalloc(&dsr, 12 * (size_t)n * sizeof(cf));
    hipMemcpy(da, a.data(), n * sizeof(cf), hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * sizeof(cf), hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * sizeof(cf), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dc, dp, dsr, n);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    std::vector<cf> hp(12 * (size_t)n), hs(12 * (size_t)n);
    hipMemcpy(hp.data(), dp, hp.size() * sizeof(cf), hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), dsr, hs.size() * sizeof(cf), hipMemcpyDeviceToHost);
    static const char* names[12] = {"cmul", "cmulc", "cfma", "cfmac", "cfnma", "cfnmac", "cadd_jd<+1>", "cadd_jd<-1>", "cadd_c", "csub_c", "cdiv_2j", "herm_downdate"};
    long bad = 0;
    for (size_t q = 0; q < hp.size(); ++q) {
        if (same(hp[q].x, hs[q].x) && same(hp[q].y, hs[q].y)) continue;
        if (bad++ < 10) {
            const size_t i = q / 12;
            printf("mismatch %s at %zu: a=(%a,%a) b=(%a,%a) c=(%a,%a): packed (%a,%a) scalar (%a,%a)\n", names[q % 12], i, a[i].x, a[i].y, b[i].x, b[i].y,
                   c[i].x, c[i].y, hp[q].x, hp[q].y, hs[q].x, hs[q].y);
        }
    }
    if (bad) { printf("FAILED: %ld mismatches\n", bad); return 1; }
    printf("ok %d operands x 12 forms\n", n);
    return 0;
}
