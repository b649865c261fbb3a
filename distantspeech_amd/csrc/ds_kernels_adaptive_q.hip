// ds_kernels_adaptive_q.hip — the 8-microphone adaptive-MVDR frame kernels with the per-bin program spread over quads of lanes
// (ds_quad.hpp: EngineQ) for gfx950.  DS_M8_ONE_THREAD=1 in the environment selects the one-thread-per-bin kernels instead (A/B runs).
#include <cstdlib>

#include "ds_kernels.hpp"
#include "ds_quad.hpp"

namespace ds {

template <int NFFT> __global__ void __launch_bounds__(NFFT / 2) ds_frames_quad_kernel(Params p) {
    typedef EngineQ<NFFT> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

template <int NFFT> hipError_t launch_frames_quad(const Params& p, int nblocks, hipStream_t stream) {
    hipLaunchKernelGGL((ds_frames_quad_kernel<NFFT>), dim3(nblocks), dim3(EngineQ<NFFT>::NT), 0, stream, p);
    return hipGetLastError();
}

template <int NFFT> static KernelInfo quad_info() {
    typedef EngineQ<NFFT> E;
    KernelInfo ki;
    ki.launch = &launch_frames_quad<NFFT>;
    ki.NP = E::NP; ki.KP = E::KP; ki.NT = E::NT;
    return ki;
}

KernelInfo lookup_adaptive_quad(int nfft, int M) {
    KernelInfo none = {nullptr, 0, 0, 0};
    const char* e = std::getenv("DS_M8_ONE_THREAD");
    if (M != 8 || (e && e[0] == '1')) return none;
    if (nfft == 256) return quad_info<256>();
    if (nfft == 512) return quad_info<512>();
    if (nfft == 1024) return quad_info<1024>();
    return none;
}

}  // namespace ds
