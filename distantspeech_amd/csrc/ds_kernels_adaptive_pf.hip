// ds_kernels_adaptive_pf.hip — instantiations of the fused frame kernel (ALGO_ADAPTIVE_PF: MVDR + McMcra post-filter gain in one pass) for gfx950.
#include "ds_kernels.hpp"

namespace ds {
KernelInfo lookup_adaptive_pf(int nfft, int M) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) return make_info<NFFT_, M_, ALGO_ADAPTIVE_PF, false>();
    DS_FOR_EACH_SHAPE_PF(X)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}
}  // namespace ds
