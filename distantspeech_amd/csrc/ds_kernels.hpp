// ds_kernels.hpp — kernel template + per-TU instantiation helpers for libdsenh.so (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "ds_core.hpp"
#if defined(DS_WITH_SHELVED)      // make SHELVED=1: the hop-pipelined and quad-lane experiments (built, bit-identical, measured slower: DESIGN.md 4.4)
#include "ds_pipe.hpp"
#endif

namespace ds {

typedef hipError_t (*launch_fn)(const Params& p, int nblocks, hipStream_t stream);

struct KernelInfo {
    launch_fn launch;
    int NP;   // float4 state planes per bin (the last one partly filled when NF % 4 != 0)
    int KP;   // padded plane length
    int NT;   // threads per block
    int NF = 0;   // state floats per bin (in memory: NF KP floats per utterance, see StateLayout)
    launch_fn launch_pipe = nullptr;   // the same frame program as a hop-level software pipeline (ds_pipe.hpp; 512-point frames): calls of several hops
    launch_fn launch_long = nullptr;   // the same frame program compiled for calls of DS_LONG_MIN_T hops or more (ds_kernels_adaptive_pf_long.hip)
};

// lookups implemented in ds_kernels_*.hip; launch == nullptr when the combination is not compiled
KernelInfo lookup_fixed(int nfft, int M);
KernelInfo lookup_adaptive_noryy(int nfft, int M);
KernelInfo lookup_adaptive_ryy(int nfft, int M);
KernelInfo lookup_adaptive_quad(int nfft, int M);   // 8 microphones, no Ryy: the per-bin program spread over quads (ds_quad.hpp); null launch = n/a
KernelInfo lookup_gsc(int nfft, int M);
KernelInfo lookup_adaptive_pf(int nfft, int M);   // ALGO_ADAPTIVE_PF: MVDR + McMcra gain in one pass (ds_kernels_adaptive_pf.hip)
launch_fn lookup_adaptive_pf_long(int nfft, int M);   // ... and its build for long calls (4 microphones, 512 points; nullptr otherwise)
KernelInfo lookup_aic(int nfft, int M);     // ALGO_AIC: the SubbandGSC chain's tail (ds_kernels_aic.hip)
KernelInfo lookup_stft(int nfft, int M, int ov = 2);      // ds_kernels_ops.hip; ov = nfft / hop: 2 or 4
KernelInfo lookup_istft(int nfft, int M, int ov = 2);
KernelInfo lookup_stft_cdr(int nfft, int M);
KernelInfo lookup_front(int nfft, int M, int L);   // DC notch + L-tap FIR bank + analysis + McCDR as one kernel (the SubbandGSC chain's front end)   // analysis + McCDR (the SubbandGSC chain's front end); M in {4, 6, 8}
KernelInfo lookup_stft_rows(int nfft);      // single-channel handles: one row per wavefront (nfft 512 / 1024), launch(p, rows, stream)
KernelInfo lookup_istft_rows(int nfft);
struct OpParams;
hipError_t launch_binop(int op, const OpParams& p, hipStream_t stream);
struct TdParams;
hipError_t launch_dcnotch(const TdParams& p, hipStream_t stream);
hipError_t launch_fir(const TdParams& p, hipStream_t stream);
hipError_t launch_pcm16_to_float(const short* pcm, float* x, long long n, int Ctot, int c0, int M, hipStream_t stream);
hipError_t launch_float_to_pcm16(const float* y, short* pcm, long long n, hipStream_t stream);
hipError_t launch_float_to_double(const float* y, double* out, long long n, hipStream_t stream);
struct TdfParams;
hipError_t launch_tdfilter(const TdfParams& p, hipStream_t stream);
// device-resident uniform counters of an operator / front-end / chain handle: cnt = {frm_cnt, ell, first_frame, aux}; advances them by
// `frames` frames (MCRA window L; mcra.py:52-56,72-74), clears first_frame, and moves aux by aux_add modulo aux_mod (FIR ping-pong parity:
// +1 mod 2 per call; WPE delay ring: +T mod ring_len)
hipError_t launch_tick(int* cnt, int frames, int L, int aux_add, int aux_mod, hipStream_t stream);
hipError_t launch_tick3(const TickArgs& a, const TickArgs& b, const TickArgs& c, hipStream_t stream);   // three handles' counters in one launch
hipError_t launch_mcspp_qavg(const float* gamma, float* out, int rows, int K, hipStream_t stream);
struct WpeParams;
hipError_t launch_wpe(const WpeParams& p, int generic, hipStream_t stream);          // C N <= 16 (ds_wpe.hpp)
hipError_t launch_wpe_wide(const WpeParams& p, int generic, hipStream_t stream);     // 16 < C N <= 80: one wavefront per bin (ds_kernels_wpe.hip)
struct Wpe64Params;
hipError_t launch_wpe64(const Wpe64Params& p, hipStream_t stream);                   // the whole recursion in double (ds_wpe64.hpp, DS_PARAM_WPE_FP64)
hipError_t launch_wpe64_init(double* state, int B, int K, long long ustride, int C, int N, hipStream_t stream);
hipError_t launch_wpe_init(float* state, int B, int K, long long ustride, int C, int N, hipStream_t stream);
hipError_t launch_wpe_fix_diag(float* state, int B, int K, long long ustride, int C, int N, hipStream_t stream);   // Im(P_ii) = 0 (an imported state)   // P = 1e-3 I, the rest zero (awpe.py:58-77)
hipError_t launch_mvdr_probe(int M, const float* bins, long long ust, int KP, int NF, int B, int K, const float* steer, long long steer_batch_stride,
                             float diag, int method, float* H, hipStream_t stream);   // DS_FIELD_H (ds_kernels_adaptive.hip)
struct FdafParams;
hipError_t launch_fdaf(const FdafParams& p, int nfft, hipStream_t stream);   // ds_kernels_fdaf.hip

#if defined(__HIPCC__)
// HOIST = false: the thread id is laundered through an empty asm at the start of every phase, so that per-thread LDS addresses are
// recomputed inside each phase instead of being hoisted out of the frame loop and kept live in VGPRs for the whole kernel.  HOIST = true
// leaves the id alone: the compiler hoists the loop-invariant address arithmetic of every phase out of the hop loop (4-microphone MVDR
// kernel: 809 -> 622 vector instructions per hop, 94 -> 127 registers — still four waves per SIMD; +8 % with 10 s per call, +4 % at one
// hop per call, profiles/r03b/hoist_ab.txt).  Which kernels can afford the registers: frames_hoist() below.
// HOIST = 1 hoists in the transform-stage phases only (stage / stage_wave), 2 in every phase.
#define DS_LAUNDER(tid) do { if constexpr (HOIST < 2) asm volatile("" : "+v"(tid)); } while (0)
#define DS_LAUNDER_STAGE(tid) do { if constexpr (HOIST < 1) asm volatile("" : "+v"(tid)); } while (0)
template <class Rg, int HOIST = 0> struct HipExec {
    Rg r;
    template <class F> __device__ __forceinline__ void phase(F f) {
        int tid = (int)threadIdx.x;
        DS_LAUNDER(tid);
        f(tid, r);
        __syncthreads();
    }
    // a phase whose LDS results are consumed by the same wavefront only (the caller guarantees it): no workgroup barrier — the LDS
    // executes one wave's accesses in order, the fence keeps the compiler from moving them across the phase boundary
    template <class F> __device__ __forceinline__ void phase_wave(F f) {
        int tid = (int)threadIdx.x;
        DS_LAUNDER(tid);
        f(tid, r);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // the transform stages' phases (workgroup barrier / wave-local): the same, with their own hoisting level
    template <class F> __device__ __forceinline__ void stage(F f) {
        int tid = (int)threadIdx.x;
        DS_LAUNDER_STAGE(tid);
        f(tid, r);
        __syncthreads();
    }
    template <class F> __device__ __forceinline__ void stage_wave(F f) {
        int tid = (int)threadIdx.x;
        DS_LAUNDER_STAGE(tid);
        f(tid, r);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // a wave-local phase in two halves: `load` reads LDS into registers, `rest` computes and writes — possibly over what other lanes of the
    // same wave have just read (in-place transform stages: the LDS executes a wave's accesses in order, and the stores depend on the
    // loads).  Here the halves simply follow each other, so the scheduler can run `rest`'s independent arithmetic under the loads' latency;
    // the CPU emulator runs `load` for every thread before `rest` for any.
    // asynchronous 16-byte copy global -> LDS with no register in between (global_load_lds_dwordx4): lane `lane` of the wavefront lands at
    // lds_piece + 16 * lane (the hardware's rule: wave-uniform base + lane x size), from its own global address.  Every piece of a tile can
    // be in flight at once; lds_load_wait() before the phase ends retires them
    __device__ __forceinline__ void lds_load16(void* lds_piece, int lane, const void* src) {
        (void)lane;
        // (cache policy 2 = nt: state streams through once per launch, like load_state())
#if defined(DS_PLAIN_STATE) || defined(DS_PLAIN_STATE_LOAD)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
#else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 2);
#endif
    }
    // s_waitcnt vmcnt(0) as the BUILTIN (0x0f70: vmcnt 0, expcnt and lgkmcnt untouched), not inline assembly: the compiler's wait-count
    // pass reads a real S_WAITCNT and knows the copies have landed.  Behind an opaque asm it kept them "pending" for the rest of the
    // kernel — every LDS access after any global store then waited for vmcnt(0), and every counted wait degraded to 0
    __device__ __forceinline__ void lds_load_wait() { __builtin_amdgcn_s_waitcnt(0x0f70); asm volatile("" ::: "memory"); }
    template <class FL, class FR> __device__ __forceinline__ void phase_wave2(FL fl, FR fr) {
        int tid = (int)threadIdx.x;
        DS_LAUNDER(tid);
        fl(tid, r);
        fr(tid, r);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    template <class FL, class FR> __device__ __forceinline__ void phase2(FL fl, FR fr) {      // the same, ending in a workgroup barrier
        int tid = (int)threadIdx.x;
        DS_LAUNDER(tid);
        fl(tid, r);
        fr(tid, r);
        __syncthreads();
    }
    __device__ __forceinline__ void sync() { __syncthreads(); }
};

// waves per SIMD the register allocator must leave room for.  The 4-microphone kernels live at the 128-VGPR step (4 waves / SIMD = 4
// resident workgroups per CU for 512-point frames: the whole 1024-utterance batch in one wave of workgroups); the GSC kernel had
// slipped to 130 VGPRs / 3 waves, so it is pinned; the MVDR kernel sits on the same edge (two more registers and a workgroup per CU
// is gone: -25 %) and is pinned too; larger arrays keep the allocator's own choice.
// The 6-microphone SubbandGSC tail (ALGO_AIC) holds three workgroups per CU by its LDS; with the packed complex products the allocator
// took 172 registers (two waves per SIMD) where 168 keep the third: pinned as well.
#ifndef DS_GSC_HOIST
#define DS_GSC_HOIST 1      // the transform stages only: 128 registers without scratch (level 2 spills 52 B); +4 % with 10 s per call, neutral at one hop
#endif                      // per call (profiles/r03f/wpe_gschoist_ab.txt: before the state went back inside the last hop this level spilled too and lost 13 %)
// frame kernels whose register budget has room for the hoisted addresses at unchanged occupancy (measured per shape with
// -Rpass-analysis=kernel-resource-usage: the GSC kernel spills, the 6- and 8-microphone kernels lose a wave per SIMD)
#ifndef DS_PF_WAVES
#define DS_PF_WAVES 4
#endif
#ifndef DS_PF_HOIST
#define DS_PF_HOIST 0      // level 2: 168 registers, three waves per SIMD = three workgroups per CU: 1024 utterances then need a second round of workgroups (20.1 us per
                           // hop against the plain MVDR kernel's 10.4); level 0 fits 128 registers without scratch: four per CU, the batch in one round
#endif
constexpr int frames_hoist(int nfft, int M, int algo, bool ryy) {
#if defined(DS_NO_HOIST)
    return 0;
#else
    if (nfft <= 512 && M <= 5 && ((algo == ALGO_ADAPTIVE && !ryy) || algo == ALGO_FIXED)) return 2;      // 5 microphones: 134 -> 168 registers, still three waves
                                                                                                          // per SIMD; +3 .. 11 % with 40 hops per call (r03f/m5_hoist_ab.txt)
    if (nfft <= 512 && M <= 4 && algo == ALGO_GSC) return DS_GSC_HOIST;
    if (nfft == 512 && M == 4 && algo == ALGO_ADAPTIVE_PF) return DS_PF_HOIST;    // 134 -> 168 registers: three waves per SIMD either way (the 256-point kernel spills 12 B at level 2)
    return 0;
#endif
}
// (with Ryy — the Python mirror's objects and TFGSC — the 4-microphone program does not fit 128 registers without scratch: three waves)
// (a 1024-point workgroup is eight waves: two per SIMD is all a CU can hold of it, whatever the register count)
constexpr int frames_min_waves(int M, int algo, bool ryy = false, int nfft = 512) {
    if (nfft >= 1024) return 1;
    return (M <= 4 && (algo == ALGO_GSC || algo == ALGO_ADAPTIVE)) ? (ryy && M == 4 ? 3 : 4) : (M == 6 && algo == ALGO_AIC) ? 3
           : (M == 4 && algo == ALGO_ADAPTIVE_PF) ? (nfft == 512 ? DS_PF_WAVES : 3) : 1;      // (the 256-point kernel spills 20 B at four)
}

template <int NFFT, int M, int ALGO, bool RYY>
__global__ void __launch_bounds__(NFFT / 2, frames_min_waves(M, ALGO, RYY, NFFT)) ds_frames_kernel(Params p) {
    typedef Engine<NFFT, M, ALGO, RYY> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg, frames_hoist(NFFT, M, ALGO, RYY)> ex;
    if constexpr (ALGO == ALGO_AIC) {                    // a stage of a chain: an earlier stage's counter advance rides in this launch
        if (blockIdx.x == 0 && threadIdx.x == 0) apply_tick(p.tick);
    }
    E::run(ex, p, (int)blockIdx.x, sh);
}

template <int NFFT, int M, int ALGO, bool RYY>
hipError_t launch_frames(const Params& p, int nblocks, hipStream_t stream) {
    typedef Engine<NFFT, M, ALGO, RYY> E;
    hipLaunchKernelGGL((ds_frames_kernel<NFFT, M, ALGO, RYY>), dim3(nblocks), dim3(E::NT), 0, stream, p);
    return hipGetLastError();
}

// The same kernel under a second name, for a translation unit that compiles the frame program differently for LONG calls (ds_kernels_adaptive_pf_long.hip:
// no SLP vectoriser, the MVDR sweep's conjugation folded into its first column — bit-identical to the default, + 1.5 % from two hops per call on,
// but - 1.6 … - 3.6 % at one hop per call for the MVDR + post-filter kernel: so that kernel exists twice and the call length picks)
template <int NFFT, int M, int ALGO, bool RYY>
__global__ void __launch_bounds__(NFFT / 2, frames_min_waves(M, ALGO, RYY, NFFT)) ds_frames_long_kernel(Params p) {
    typedef Engine<NFFT, M, ALGO, RYY> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg, frames_hoist(NFFT, M, ALGO, RYY)> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}
template <int NFFT, int M, int ALGO, bool RYY>
hipError_t launch_frames_long(const Params& p, int nblocks, hipStream_t stream) {
    typedef Engine<NFFT, M, ALGO, RYY> E;
    hipLaunchKernelGGL((ds_frames_long_kernel<NFFT, M, ALGO, RYY>), dim3(nblocks), dim3(E::NT), 0, stream, p);
    return hipGetLastError();
}
#ifndef DS_LONG_MIN_T
#define DS_LONG_MIN_T 2
#endif

#if defined(DS_WITH_SHELVED)
// the hop-pipelined form of the same program (ds_pipe.hpp): same state, same arguments, same results bit for bit
template <int NFFT, int M, int ALGO, bool RYY>
__global__ void __launch_bounds__(NFFT / 2, frames_min_waves(M, ALGO)) ds_frames_pipe_kernel(Params p) {
    typedef PipeEngine<NFFT, M, ALGO, RYY> E;
    __shared__ typename E::Sh sh;
    HipExec<typename E::Rg> ex;
    E::run(ex, p, (int)blockIdx.x, sh);
}

template <int NFFT, int M, int ALGO, bool RYY>
hipError_t launch_frames_pipe(const Params& p, int nblocks, hipStream_t stream) {
    typedef PipeEngine<NFFT, M, ALGO, RYY> E;
    hipLaunchKernelGGL((ds_frames_pipe_kernel<NFFT, M, ALGO, RYY>), dim3(nblocks), dim3(E::NT), 0, stream, p);
    return hipGetLastError();
}
#endif

template <int NFFT, int M, int ALGO, bool RYY> KernelInfo make_info() {
    typedef Engine<NFFT, M, ALGO, RYY> E;
    KernelInfo ki;
    ki.launch = &launch_frames<NFFT, M, ALGO, RYY>;
    ki.NP = E::NP; ki.NF = E::SL::NF; ki.KP = E::KP; ki.NT = E::NT;
#if defined(DS_WITH_SHELVED)
    if constexpr (NFFT == 512 && (ALGO == ALGO_FIXED || ALGO == ALGO_ADAPTIVE || ALGO == ALGO_GSC) && M <= 4)
        ki.launch_pipe = &launch_frames_pipe<NFFT, M, ALGO, RYY>;
#endif
    return ki;
}

#define DS_FOR_EACH_SHAPE(X) \
    X(256, 2) X(256, 3) X(256, 4) X(256, 5) X(256, 6) X(256, 8) \
    X(512, 2) X(512, 3) X(512, 4) X(512, 5) X(512, 6) X(512, 8) \
    X(1024, 2) X(1024, 3) X(1024, 4) X(1024, 5) X(1024, 6) X(1024, 8)
// MVDR + post-filter in one thread per bin: 2 .. 6 and (round 6) 8 microphones.  At 8 the two programs hold 141 state floats per lane: 344 - 348
// registers = one wave per SIMD at 256 / 512 points, 380 B of scratch at 1024 points (a 1024-point workgroup is two waves per SIMD whatever
// it declares) — built so that the handle takes every shape the plain MVDR handle takes; the chain DS_ALGO_WPE_MVDR keeps running the two
// programs as two operators at two waves each
#define DS_FOR_EACH_SHAPE_PF(X) \
    X(256, 2) X(256, 3) X(256, 4) X(256, 5) X(256, 6) X(256, 8) \
    X(512, 2) X(512, 3) X(512, 4) X(512, 5) X(512, 6) X(512, 8) \
    X(1024, 2) X(1024, 3) X(1024, 4) X(1024, 5) X(1024, 6) X(1024, 8)
#endif

}  // namespace ds
