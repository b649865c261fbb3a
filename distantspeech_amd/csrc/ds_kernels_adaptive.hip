// ds_kernels_adaptive.hip — instantiations of the fused frame kernel (ALGO_ADAPTIVE, RYY=false) for gfx950.
#include "ds_kernels.hpp"

namespace ds {
KernelInfo lookup_adaptive_noryy(int nfft, int M) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) return make_info<NFFT_, M_, ALGO_ADAPTIVE, false>();
    DS_FOR_EACH_SHAPE(X)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}

// DS_FIELD_H: the MVDR weights the frame kernel applies, from the frame kernel's own solve.  The kernel never forms w = A^-1 a / (a^H A^-1 a):
// it computes Y = w^H z as (u^H t) / (u^H u) in one fused Cholesky sweep (mvdr_output).  This read-only probe runs THAT function on the
// handle's packed Rvv planes with the unit frames z = e_m: Y_m = conj(w_m).  One thread per (utterance, bin); no state is written.
template <int M>
__global__ void __launch_bounds__(256) ds_mvdr_probe_kernel(const float* bins, long long ust, int KP, int NF, int B, int K, const cf* steer,
                                                            long long steer_batch_stride, float diag, int method, float* H) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= (long long)B * K) return;
    const int b = (int)(g / K), k = (int)(g - (long long)b * K);
    const int NPF = NF / 4, RT = NF % 4;
    const float* u = bins + (long long)b * ust;
    auto at = [&](int f) { return f < 4 * NPF ? u[((long long)(f / 4) * KP + k) * 4 + (f % 4)] : u[(long long)NPF * KP * 4 + (long long)k * RT + (f - 4 * NPF)]; };
    float d[M], o[M * (M - 1) + 2];
    cf a[M];
#pragma unroll
    for (int i = 0; i < M; ++i) { d[i] = at(i); a[i] = steer[(long long)b * steer_batch_stride + (long long)k * M + i]; }
#pragma unroll
    for (int i = 0; i < M * (M - 1); ++i) o[i] = at(M + i);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        cf z[M];
#pragma unroll
        for (int i = 0; i < M; ++i) z[i] = mk(i == m ? 1.0f : 0.0f, 0.0f);
        cf y;
        if (method == METHOD_SRC) y = cmulc(z[0], a[0]);                              // adaptive_bin's branches, word for word
        else if (method == METHOD_DS) { y = mk(0.0f, 0.0f); for (int i = 0; i < M; ++i) y = cfmac(y, z[i], a[i]); y = cscale(y, 1.0f / M); }
        else y = mvdr_output<M>(d, o, diag, a, z);
        H[(g * M + m) * 2] = y.x; H[(g * M + m) * 2 + 1] = -y.y;
    }
}

hipError_t launch_mvdr_probe(int M, const float* bins, long long ust, int KP, int NF, int B, int K, const float* steer, long long steer_batch_stride,
                             float diag, int method, float* H, hipStream_t stream) {
    const unsigned blocks = (unsigned)(((long long)B * K + 255) / 256);
#define X(M_) if (M == M_) { hipLaunchKernelGGL(ds_mvdr_probe_kernel<M_>, dim3(blocks), dim3(256), 0, stream, bins, ust, KP, NF, B, K, \
                              reinterpret_cast<const cf*>(steer), steer_batch_stride, diag, method, H); return hipGetLastError(); }
    X(2) X(3) X(4) X(5) X(6) X(8)
#undef X
    return hipErrorInvalidValue;
}
}  // namespace ds
