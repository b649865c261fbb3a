// ds_kernels_wpe.hip — the wide-tap RLS-WPE kernels (ds_wpe_wide.hpp) for gfx950: one wavefront per (utterance, bin), 16 < C N <= 80.
// Two wavefronts per SIMD (eight per CU): the 80 x 80 matrix is 200 registers per lane, the tile 13 KB of LDS per wavefront.
#include "ds_kernels.hpp"
#include "ds_wpe_wide.hpp"
#include "ds_wpe64.hpp"

namespace ds {

template <int CNP, int NCH, int CT = 0, int NTAPS = 0>
__global__ void __launch_bounds__(WPEW_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) ds_wpe_wide_kernel(WpeParams p) {
    typedef WpeWideEngine<CNP, NCH, CT, NTAPS> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

// the reference's notebook shape (4 channels x 20 taps, example/wpe.ipynb cell 2) and SURVEY 8(d)'s cfg4 sizing (8 x 10) with the channel
// and tap counts as compile-time constants; any other 16 < C N <= 80 through the run-time-shape kernel of its padded size (32 / 64 / 80).
// generic != 0 (DS_WPE_GENERIC=1, read once at ds_create): every shape through the run-time-shape kernels (A/B and tests)
hipError_t launch_wpe_wide(const WpeParams& p, int generic, hipStream_t stream) {
    const int CN = p.C * p.N;
    if (CN <= WPE_CNMAX || CN > WPEW_CNMAX || p.C > WPE_CMAX) return hipErrorInvalidValue;
    const long long blocks = (long long)p.B * p.K;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define DS_WPEW(...) do { hipLaunchKernelGGL((ds_wpe_wide_kernel<__VA_ARGS__>), dim3((unsigned)blocks), dim3(WPEW_NT), 0, stream, p); return hipGetLastError(); } while (0)
    if (!(generic & 1)) {
        if (p.C == 4 && p.N == 20) DS_WPEW(80, 2, 4, 20);
        if (p.C == 8 && p.N == 10) DS_WPEW(80, 2, 8, 10);
    }
    if (CN <= 32) DS_WPEW(32, 1);
    if (CN <= 64) DS_WPEW(64, 2);
    DS_WPEW(80, 2);
#undef DS_WPEW
}

// the double-precision recursion (ds_wpe64.hpp, DS_PARAM_WPE_FP64): one workgroup per (utterance, bin), the whole matrix in LDS
template <int CNP> __global__ void __launch_bounds__(WPE64_NT) ds_wpe64_kernel(Wpe64Params p) {
    typedef Wpe64Engine<CNP> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
hipError_t launch_wpe64(const Wpe64Params& p, hipStream_t stream) {
    const int CN = p.w.C * p.w.N;
    const long long blocks = (long long)p.w.B * p.w.K;
    if (CN < 1 || CN > WPEW_CNMAX || p.w.C > WPE_CMAX || blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (CN <= 16) hipLaunchKernelGGL(ds_wpe64_kernel<16>, dim3((unsigned)blocks), dim3(WPE64_NT), 0, stream, p);
    else if (CN <= 32) hipLaunchKernelGGL(ds_wpe64_kernel<32>, dim3((unsigned)blocks), dim3(WPE64_NT), 0, stream, p);
    else hipLaunchKernelGGL(ds_wpe64_kernel<80>, dim3((unsigned)blocks), dim3(WPE64_NT), 0, stream, p);
    return hipGetLastError();
}
// its initial state: P = 1e-3 I, everything else zero (awpe.py:58-77)
__global__ void __launch_bounds__(256) ds_wpe64_init_kernel(double* state, long long n, int K, long long ustride, long long SB, int CN) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int i = (int)(g % CN);
    const long long bk = g / CN, b = bk / K, k = bk - b * K;
    state[b * ustride + k * SB + 2 * ((long long)i * CN + i)] = 1e-3;
}
hipError_t launch_wpe64_init(double* state, int B, int K, long long ustride, int C, int N, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(state, 0, (size_t)B * (size_t)ustride * sizeof(double), stream);
    if (e != hipSuccess) return e;
    const long long n = (long long)B * K * C * N;
    hipLaunchKernelGGL(ds_wpe64_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, state, n, K, ustride, wpe64_bin_doubles(C, N), C * N);
    return hipGetLastError();
}

// P = 1e-3 I (the diagonal words of the packed triangle, awpe.py:69-73) on a zeroed state: one thread per (utterance, bin, tap)
template <bool INIT> __global__ void __launch_bounds__(256) ds_wpe_init_kernel(float* state, long long n, int K, long long ustride, int SB, int CN) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int i = (int)(g % CN);
    const long long bk = g / CN, b = bk / K, k = bk - b * K;
    float* pii = state + b * ustride + k * SB + 2 * (wpew_words(i) + i);
    if (INIT) pii[0] = 1e-3f; else pii[1] = 0.0f;       // the diagonal word of the packed triangle: P_ii = 1e-3 / Im(P_ii) = 0
}
hipError_t launch_wpe_init(float* state, int B, int K, long long ustride, int C, int N, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(state, 0, (size_t)B * (size_t)ustride * sizeof(float), stream);
    if (e != hipSuccess) return e;
    const long long n = (long long)B * K * C * N;
    hipLaunchKernelGGL(ds_wpe_init_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, state, n, K, ustride, wpe_bin_floats(C, N), C * N);
    return hipGetLastError();
}
hipError_t launch_wpe_fix_diag(float* state, int B, int K, long long ustride, int C, int N, hipStream_t stream) {
    const long long n = (long long)B * K * C * N;
    hipLaunchKernelGGL(ds_wpe_init_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, state, n, K, ustride, wpe_bin_floats(C, N), C * N);
    return hipGetLastError();
}

}  // namespace ds
