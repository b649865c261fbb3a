// ds_kernels_adaptive_q.hip — the 8-microphone adaptive-MVDR frame kernels with the per-bin program spread over quads of lanes
// (ds_quad.hpp: EngineQ) for gfx950.  Opt-in with DS_M8_QUAD=1 in the environment: the kernels carry no scratch (200 .. 226 VGPRs against
// 256 + 172 .. 828 B of scratch per lane) and give the one-thread kernels' numbers bit for bit, but the frame program at 8 microphones is
// bound by VALU issue, not by the spills, and the quad form issues about twice the instructions (broadcasts, masked padding, values
// every lane recomputes): measured 54.0 vs 54.9 us per 1024-utterance hop at 512 points, 128.5 vs 104.1 us at 1024 points, 2557 vs 1916 us
// at 40 hops per call (profiles/r02b/m8_quad_ab.txt).  The one-thread kernels therefore stay the default.
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
    ki.NP = E::NP; ki.KP = E::KP; ki.NT = E::NT; ki.NF = E::EB::SL::NF;       // the state layout is the one-thread kernel's
    return ki;
}

KernelInfo lookup_adaptive_quad(int nfft, int M) {
    KernelInfo none = {nullptr, 0, 0, 0};
    const char* e = std::getenv("DS_M8_QUAD");
    if (M != 8 || !(e && e[0] == '1')) return none;
    if (nfft == 256) return quad_info<256>();
    if (nfft == 512) return quad_info<512>();
    if (nfft == 1024) return quad_info<1024>();
    return none;
}

}  // namespace ds
