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
}  // namespace ds
