// ds_kernels_fdaf.hip — instantiations of the overlap-save FDAF block program (ds_fdaf.hpp) for gfx950:
// n_fft in {128, 256, 512, 1024} (filter_len 64..512) x channel capacity {1, 4, 8}.
#include "ds_kernels.hpp"
#include "ds_fdaf.hpp"

namespace ds {

template <int NFFT, int CMAX>
__global__ void __launch_bounds__(NFFT / 2) ds_fdaf_kernel(FdafParams p) {
    typedef FdafEngine<NFFT, CMAX> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

hipError_t launch_fdaf(const FdafParams& p, int nfft, hipStream_t stream) {
#define X(N_, C_)                                                                                            \
    if (nfft == N_ && p.C <= C_ && !(p.two_path && C_ < 2)) {   /* two_path runs the foreground output as a second transform channel */                                                                           \
        hipLaunchKernelGGL((ds_fdaf_kernel<N_, C_>), dim3(p.B), dim3(N_ / 2), 0, stream, p);                   \
        return hipGetLastError();                                                                              \
    }
    X(128, 1) X(128, 4) X(128, 8) X(256, 1) X(256, 4) X(256, 8) X(512, 1) X(512, 4) X(512, 8) X(1024, 1) X(1024, 4) X(1024, 8)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace ds
