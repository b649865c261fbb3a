// ds_kernels_aic.hip — instantiations of the fused frame kernel with the SubbandGSC chain's tail as its per-bin program (ALGO_AIC:
// re-analysis of the blocking-matrix outputs -> 2-tap multi-channel subband NLMS canceller -> synthesis) for gfx950.
#include "ds_kernels.hpp"

namespace ds {
KernelInfo lookup_aic(int nfft, int M) {
#define X(NFFT_, M_) if (nfft == NFFT_ && M == M_) return make_info<NFFT_, M_, ALGO_AIC, false>();
    DS_FOR_EACH_SHAPE(X)
#undef X
    KernelInfo none = {nullptr, 0, 0, 0};
    return none;
}
}  // namespace ds
