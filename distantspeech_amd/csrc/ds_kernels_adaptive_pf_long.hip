// ds_kernels_adaptive_pf_long.hip — the MVDR + McMcra post-filter frame kernel (4 microphones, 512 points: the bench workload `mvdr_pf`) once more,
// built for calls of DS_LONG_MIN_T hops or more: this unit is compiled with -fno-slp-vectorize (Makefile: NOSLP) and with the MVDR sweep's
// conjugation folded into its first column (DS_PF_LONG -> adaptive_bin<.., CF = true>).  Both leave every result bit for bit as it is (packed or
// not, IEEE operations per word; exact negations) — the GPU tests hold one long call against hop-by-hop calls.  Measured against the default build
// (profiles/r05a/pf_long_sweep.txt): + 1.2 % at 2 hops per call, + 1.5 % from 8 hops to 625; at one hop per call (HBM-bound) the same two changes
// cost 1.6 % and 3.6 %, so the default build stays there.
#define DS_PF_LONG 1
#include "ds_kernels.hpp"

namespace ds {
launch_fn lookup_adaptive_pf_long(int nfft, int M) {
    if (nfft == 512 && M == 4) return &launch_frames_long<512, 4, ALGO_ADAPTIVE_PF, false>;
    return nullptr;
}
}  // namespace ds
