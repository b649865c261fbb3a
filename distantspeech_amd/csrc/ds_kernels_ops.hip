// ds_kernels_ops.hip — stand-alone STFT / ISTFT kernels and the frame-level (utterance, bin) operator
// kernel (MCRA, McMcra, NsOmlsaMulti, subband LMS / RLS) for gfx950.
#include "ds_kernels.hpp"
#include "ds_ops.hpp"
#include "ds_tdfilter.hpp"

namespace ds {

template <int NFFT, int M> __global__ void __launch_bounds__(NFFT / 2) ds_stft_kernel(Params p) {
    typedef StftEngine<NFFT, M> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

template <int NFFT, int M> __global__ void __launch_bounds__(NFFT / 2) ds_istft_kernel(Params p) {
    typedef IstftEngine<NFFT, M> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

template <int NFFT, int M> hipError_t launch_stft(const Params& p, int nblocks, hipStream_t stream) {
    hipLaunchKernelGGL((ds_stft_kernel<NFFT, M>), dim3(nblocks), dim3(NFFT / 2), 0, stream, p);
    return hipGetLastError();
}
template <int NFFT, int M> hipError_t launch_istft(const Params& p, int nblocks, hipStream_t stream) {
    hipLaunchKernelGGL((ds_istft_kernel<NFFT, M>), dim3(nblocks), dim3(NFFT / 2), 0, stream, p);
    return hipGetLastError();
}

#define DS_FOR_EACH_TSHAPE(X) \
    X(256, 1) X(256, 2) X(256, 4) X(256, 6) X(256, 8) \
    X(512, 1) X(512, 2) X(512, 4) X(512, 6) X(512, 8) \
    X(1024, 1) X(1024, 2) X(1024, 4) X(1024, 6) X(1024, 8)

KernelInfo lookup_stft(int nfft, int M) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) { KernelInfo ki = {&launch_stft<NFFT_, M_>, 0, (NFFT_ / 2 + 4) & ~3, NFFT_ / 2}; return ki; }
    DS_FOR_EACH_TSHAPE(X)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}
KernelInfo lookup_istft(int nfft, int M) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) { KernelInfo ki = {&launch_istft<NFFT_, M_>, 0, (NFFT_ / 2 + 4) & ~3, NFFT_ / 2}; return ki; }
    DS_FOR_EACH_TSHAPE(X)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}

template <int OP, int M> __global__ void __launch_bounds__(256) ds_binop_kernel(OpParams p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(i / p.KP), k = (int)(i - (long long)b * p.KP);
    if (b >= p.B || k >= p.K) return;
    run_op_t<OP, M>(p, b, k);
}

hipError_t launch_binop(int op, const OpParams& p, hipStream_t stream) {
    const long long total = (long long)p.B * p.KP;
    const int blocks = (int)((total + 255) / 256);
#define X(OP_, M_)                                                                                         \
    if (op == OP_ && (!op_is_matrix(OP_) || p.M == M_)) {                                                    \
        hipLaunchKernelGGL((ds_binop_kernel<OP_, M_>), dim3(blocks), dim3(256), 0, stream, p);               \
        return hipGetLastError();                                                                            \
    }
    DS_FOR_EACH_OP(X)
#undef X
    return hipErrorInvalidValue;
}

__global__ void __launch_bounds__(64) ds_dcnotch_kernel(TdParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.B * p.M) td_dcnotch(p, i / p.M, i % p.M);
}
__global__ void __launch_bounds__(256) ds_fir_kernel(TdParams p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)p.B * p.n) td_fir(p, (int)(i / p.n), (int)(i % p.n));
}
__global__ void __launch_bounds__(256) ds_fir_cache_kernel(TdParams p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)p.B * (p.L - 1)) td_fir_cache(p, (int)(i / (p.L - 1)), (int)(i % (p.L - 1)));
}

hipError_t launch_dcnotch(const TdParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(ds_dcnotch_kernel, dim3((p.B * p.M + 63) / 64), dim3(64), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_fir(const TdParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(ds_fir_kernel, dim3((unsigned)(((long long)p.B * p.n + 255) / 256)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(ds_fir_cache_kernel, dim3((unsigned)(((long long)p.B * (p.L - 1) + 255) / 256)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

__global__ void __launch_bounds__(TDF_NT) ds_tdfilter_kernel(TdfParams p) {
    __shared__ TdfShared sh;
    HipExec<TdfRegs> ex;
    TdfEngine::run(ex, p, (int)blockIdx.x, sh);
}
hipError_t launch_tdfilter(const TdfParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(ds_tdfilter_kernel, dim3(p.B), dim3(TDF_NT), 0, stream, p);
    return hipGetLastError();
}

template <int LPB> __global__ void __launch_bounds__(WPE_NT) ds_wpe_kernel(WpeParams p) {
    typedef WpeEngine<LPB> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
hipError_t launch_wpe(const WpeParams& p, hipStream_t stream) {
    const int lpb = wpe_lanes_per_bin(p.C * p.N), bpw = WPE_NT / lpb;
    const unsigned blocks = (unsigned)(((long long)p.B * p.K + bpw - 1) / bpw);
    if (lpb == 4) hipLaunchKernelGGL(ds_wpe_kernel<4>, dim3(blocks), dim3(WPE_NT), 0, stream, p);
    else if (lpb == 8) hipLaunchKernelGGL(ds_wpe_kernel<8>, dim3(blocks), dim3(WPE_NT), 0, stream, p);
    else hipLaunchKernelGGL(ds_wpe_kernel<16>, dim3(blocks), dim3(WPE_NT), 0, stream, p);
    return hipGetLastError();
}

// realtime wire format (realtime/realtime_processing.py:119-133): int16 LE interleaved [L][C_total] -> float32 / 32768,
// channels [c0, c0 + M) -> x [B][L][M]; enhanced float -> (y * 32768) truncated to int16
__global__ void __launch_bounds__(256) ds_pcm16_to_float_kernel(const short* pcm, float* x, long long n, int Ctot, int c0, int M) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over B * L * M
    if (i >= n) return;
    const long long s = i / M;
    const int m = (int)(i - s * M);
    x[i] = (float)pcm[s * Ctot + c0 + m] / 32768.0f;
}
__global__ void __launch_bounds__(256) ds_float_to_pcm16_kernel(const float* y, short* pcm, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = y[i] * 32768.0f;
    v = fminf(fmaxf(v, -32768.0f), 32767.0f);                                   // astype('<i2') would wrap; saturate instead
    pcm[i] = (short)(int)v;                                                     // truncation toward zero like astype
}
hipError_t launch_pcm16_to_float(const short* pcm, float* x, long long n, int Ctot, int c0, int M, hipStream_t stream) {
    hipLaunchKernelGGL(ds_pcm16_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, pcm, x, n, Ctot, c0, M);
    return hipGetLastError();
}
hipError_t launch_float_to_pcm16(const float* y, short* pcm, long long n, hipStream_t stream) {
    hipLaunchKernelGGL(ds_float_to_pcm16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, y, pcm, n);
    return hipGetLastError();
}

}  // namespace ds
