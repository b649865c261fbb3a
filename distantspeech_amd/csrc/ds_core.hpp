// ds_core.hpp — single-source block program of the fused per-frame enhancement kernel.
//
// One workgroup owns one utterance and walks its frames in order:
//   hop of M-channel samples -> LDS -> windowed real FFT (packed N/2-point complex Stockham, LDS)
//   -> one thread per frequency bin: MCRA / covariance recursion / Hermitian solve / weights
//   -> inverse packed real FFT -> windowed overlap-add -> hop of enhanced samples.
// Per-bin state (covariances, MCRA trackers) lives in registers for the whole call and in HBM
// between calls as float4 planes [plane][bin] so that every state access is a 16-byte-per-lane
// coalesced load/store.
//
// The program is written against an "Exec" policy: ex.phase(f) runs f(tid, regs) for every
// thread of the block and then synchronises.  dsenh.hip instantiates it with a HIP policy
// (f runs once per GPU thread, then __syncthreads()).  tests/emul/ instantiates it with a serial
// CPU policy purely to unit-test the index arithmetic without a GPU; that build is test
// infrastructure and is never loaded by the product package.
//
// Reference semantics (paths relative to /root/reference/DistantSpeech):
//   STFT/ISTFT            transform/transform.py:407-481
//   MCRA                  noise_estimation/mcra.py:27-77, NoiseEstimationBase.py:56-60
//   adaptive MVDR         beamformer/adaptivebeamformer.py:44-128, beamformer/beamformer.py:306-336
//   fixed beamformer      beamformer/fixedbeamformer.py:147-207
//   FD-GSC + McMcra gain  beamformer/GSC.py:174-294, noise_estimation/mc_mcra.py:91-224
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DS_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define DS_HD inline
#endif

namespace ds {

#if defined(__HIPCC__)
typedef float4 vec4;
#else
struct alignas(16) vec4 { float x, y, z, w; };
#endif

// keeps the compiler from forwarding values across it (used around a deliberate round trip through LDS); no instruction emitted
#if defined(__HIP_DEVICE_COMPILE__)
#define DS_COMPILER_FENCE() asm volatile("" ::: "memory")
#if defined(DS_NO_SETPRIO)
#define DS_SETPRIO(n) ((void)0)
#else
#define DS_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
#define DS_GRID_BLOCKS() ((int)gridDim.x)
#define DS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// orders arithmetic the optimiser would otherwise hoist: the values come out of an (empty) volatile asm, and volatile asms keep their
// program order — what is computed from a pinned value starts after everything that feeds an earlier pin.  No instruction emitted
#define DS_ASSUME(c) __builtin_assume(c)
#define DS_PIN(a) asm volatile("" : "+v"(a))
#define DS_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#else
#define DS_ASSUME(c) do { if (!(c)) __builtin_trap(); } while (0)      // the CPU emulator checks what the device build assumes
#define DS_PIN(a) ((void)0)
#define DS_PIN2(a, b) ((void)0)
#define DS_COMPILER_FENCE() ((void)0)
#define DS_SCHED_FENCE() ((void)0)
#define DS_SETPRIO(n) ((void)0)
#define DS_GRID_BLOCKS() 0
#endif

struct cf { float x, y; };

// A coefficient the caller wrote as a decimal (0.98, 0.998) arrives as a float through ds_config; where the recursion that uses it runs in
// double (the DC notch, the RLS-WPE recursion's double mode) the double the caller MEANT is the shortest decimal that rounds to that float
// (<= 7 significant digits), which is what the reference's Python float is: 0.98f = 0.98 (1 + 1.9e-8) -> 0.98.  Host side only.
inline double decimal_double(float v) {
    char buf[32];
    snprintf(buf, sizeof buf, "%.7g", (double)v);
    return strtod(buf, nullptr);
}

// All floating-point contraction is explicit: the library is compiled with -ffp-contract=off and every
// fused multiply-add below is written as fma_(), so a value's rounding never depends on which copy of
// an unrolled / peeled loop the compiler happened to emit (chunked calls == one long call, bit for bit).
DS_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DS_HD cf mk(float a, float b) { cf r; r.x = a; r.y = b; return r; }
// Reciprocal, reciprocal square root, 2^x and log2 x as the hardware's one-instruction forms (1 ulp) on the device, where the correctly
// rounded division / square root / libm call is 10-20 instructions each: used in the post-filter arithmetic of the McMcra / GSC bin
// program (a dozen of them per bin and frame), whose parity bar is 1e-4 against a float64 reference.  The CPU build keeps the exact forms.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_EXACT_DIV)
DS_HD float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
DS_HD float rsq_(float x) { return __builtin_amdgcn_rsqf(x); }
DS_HD float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
DS_HD float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
DS_HD float exp2_(float x) { return __builtin_amdgcn_exp2f(x); }
DS_HD float log2_(float x) { return __builtin_amdgcn_logf(x); }
DS_HD float exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
#else
DS_HD float rcp_(float x) { return 1.0f / x; }
DS_HD float rsq_(float x) { return 1.0f / sqrtf(x); }
DS_HD float sqrt_(float x) { return sqrtf(x); }
DS_HD float div_(float a, float b) { return a / b; }
DS_HD float exp2_(float x) { return exp2f(x); }
DS_HD float log2_(float x) { return log2f(x); }
DS_HD float exp_(float x) { return expf(x); }
#endif
DS_HD cf cadd(cf a, cf b) { return mk(a.x + b.x, a.y + b.y); }
DS_HD cf csub(cf a, cf b) { return mk(a.x - b.x, a.y - b.y); }
// The complex products, as the scalar expressions that DEFINE their rounding (every product and every fused multiply-add below is one
// rounding, in this nesting) ...
DS_HD cf cmul_s(cf a, cf b) { return mk(fma_(a.x, b.x, -(a.y * b.y)), fma_(a.x, b.y, a.y * b.x)); }
DS_HD cf cmulc_s(cf a, cf b) { return mk(fma_(a.x, b.x, a.y * b.y), fma_(a.y, b.x, -(a.x * b.y))); }   // a * conj(b)
DS_HD cf cfma_s(cf acc, cf a, cf b) {        // acc + a * b
    return mk(fma_(a.x, b.x, fma_(-a.y, b.y, acc.x)), fma_(a.x, b.y, fma_(a.y, b.x, acc.y)));
}
DS_HD cf cfmac_s(cf acc, cf a, cf b) {       // acc + a * conj(b)
    return mk(fma_(a.x, b.x, fma_(a.y, b.y, acc.x)), fma_(a.y, b.x, fma_(-a.x, b.y, acc.y)));
}
DS_HD cf cfnma_s(cf acc, cf a, cf b) {       // acc - a * b
    return mk(fma_(-a.x, b.x, fma_(a.y, b.y, acc.x)), fma_(-a.x, b.y, fma_(-a.y, b.x, acc.y)));
}
DS_HD cf cfnmac_s(cf acc, cf a, cf b) {      // acc - a * conj(b)
    return mk(fma_(-a.x, b.x, fma_(-a.y, b.y, acc.x)), fma_(-a.y, b.x, fma_(a.x, b.y, acc.y)));
}
// P / lambda - (g_i conj(g_j)) dls, the element update of the RLS-WPE recursion (ds_wpe.hpp): every product of g_i conj(g_j) is rounded on its
// own and the two of a sum are then added, so that the element lane j computes for (j, i) is the exact conjugate of this one
DS_HD cf herm_downdate_s(cf P, cf gi, cf gj, float lam_inv, float dls) {
    const float tx = gi.x * gj.x + gi.y * gj.y, ty = gi.y * gj.x - gi.x * gj.y;
    return mk(fma_(-tx, dls, P.x * lam_inv), fma_(-ty, dls, P.y * lam_inv));
}
// the same update with the scale folded into the vectors: P / lambda - h_i conj(h_j), h = g sqrt(1 / (den lambda)) (ds_wpe_wide.hpp).  Same
// symmetry: both products of a sum are rounded on their own, so (j, i) comes out as the exact conjugate of (i, j)
DS_HD cf herm_downdate_h_s(cf P, cf hi, cf hj, float lam_inv) {
    const float tx = hi.x * hj.x + hi.y * hj.y, ty = hi.y * hj.x - hi.x * hj.y;
    return mk(fma_(P.x, lam_inv, -tx), fma_(P.y, lam_inv, -ty));
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_SCALAR_COMPLEX)
// ... and on the device as TWO packed instructions each (v_pk_mul_f32 / v_pk_fma_f32): the half selects (op_sel / op_sel_hi: which half of
// a source feeds the low / the high result) and the per-half negations (neg_lo / neg_hi) do the swaps and sign changes of a complex product
// for free.  The compiler's own vectoriser packs the scalar expressions too, but pays a v_mov / v_xor for every swap or one-sided negation
// (38 instructions where these forms take 26 in a mix of six a*conj(b) and six multiply-adds).  Same products, same nesting, same single
// roundings as the scalar forms: bit-identical (tests/test_gpu_ops.py::test_packed_complex_helpers_equal_their_scalar_definitions).
typedef float cf2_t __attribute__((ext_vector_type(2)));
DS_HD cf2_t cf_pk(cf a) { return __builtin_bit_cast(cf2_t, a); }
DS_HD cf pk_cf(cf2_t a) { return __builtin_bit_cast(cf, a); }
DS_HD cf cmul(cf a, cf b) {
    cf2_t t, r;                                  // t = (-(a.y b.y), a.y b.x);  r = (a.x b.x + t.lo, a.x b.y + t.hi)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(cf_pk(a)), "v"(cf_pk(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(t));
    return pk_cf(r);
}
DS_HD cf cmulc(cf a, cf b) {
    cf2_t t, r;                                  // t = (a.y b.y, -(a.x b.y));  r = (a.x b.x + t.lo, a.y b.x + t.hi)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(cf_pk(a)), "v"(cf_pk(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(t));
    return pk_cf(r);
}
DS_HD cf cfma(cf acc, cf a, cf b) {
    cf2_t t, r;                                  // t = (-a.y b.y + acc.x, a.y b.x + acc.y);  r = (a.x b.x + t.lo, a.x b.y + t.hi)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(cf_pk(acc)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(t));
    return pk_cf(r);
}
DS_HD cf cfmac(cf acc, cf a, cf b) {
    cf2_t t, r;                                  // t = (a.y b.y + acc.x, -a.x b.y + acc.y);  r = (a.x b.x + t.lo, a.y b.x + t.hi)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(t) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(cf_pk(acc)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(t));
    return pk_cf(r);
}
DS_HD cf cfnma(cf acc, cf a, cf b) {
    cf2_t t, r;                                  // t = (a.y b.y + acc.x, -a.y b.x + acc.y);  r = (-a.x b.x + t.lo, -a.x b.y + t.hi)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(t) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(cf_pk(acc)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(t));
    return pk_cf(r);
}
DS_HD cf cfnmac(cf acc, cf a, cf b) {
    cf2_t t, r;                                  // t = (-a.y b.y + acc.x, a.x b.y + acc.y);  r = (-a.x b.x + t.lo, -a.y b.x + t.hi)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(t) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(cf_pk(acc)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(t));
    return pk_cf(r);
}
// conj(acc) - a conj(b) and conj(a) s: the conjugations as the packed instructions' own per-half negations (cfnmac / cscale of a conjugated
// operand cost a v_xor and a v_mov per use to build the conjugate in a register pair first)
DS_HD cf cfnmac_ca(cf acc, cf a, cf b) {
    cf2_t t, r;                                  // t = (-a.y b.y + acc.x, a.x b.y - acc.y);  r = (-a.x b.x + t.lo, -a.y b.x + t.hi)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[0,0,1]" : "=v"(t) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(cf_pk(acc)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)), "v"(t));
    return pk_cf(r);
}
DS_HD cf cscale_c(cf a, float s) {
    cf2_t r;
    const cf2_t sv = {s, s};
    asm("v_pk_mul_f32 %0, %1, %2 neg_hi:[1,0]" : "=v"(r) : "v"(cf_pk(a)), "v"(sv));
    return pk_cf(r);
}
DS_HD cf herm_downdate(cf P, cf gi, cf gj, float lam_inv, float dls) {     // five packed instructions for the eleven of the scalar form
    cf2_t p1, p2, t, q, r;
    const cf2_t sc = {lam_inv, dls};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p1) : "v"(cf_pk(gi)), "v"(cf_pk(gj)));      // (gi.x gj.x, gi.y gj.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(p2) : "v"(cf_pk(gi)), "v"(cf_pk(gj)));      // (gi.y gj.y, gi.x gj.y)
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(t) : "v"(p1), "v"(p2));                                     // (tx, ty)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(q) : "v"(cf_pk(P)), "v"(sc));               // P lam_inv
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(r) : "v"(t), "v"(sc), "v"(q));   // -t dls + q
    return pk_cf(r);
}
DS_HD cf herm_downdate_h(cf P, cf hi, cf hj, float lam_inv) {               // four packed instructions
    cf2_t p1, p2, t, r;
    const cf2_t sc = {lam_inv, lam_inv};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p1) : "v"(cf_pk(hi)), "v"(cf_pk(hj)));      // (hi.x hj.x, hi.y hj.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(p2) : "v"(cf_pk(hi)), "v"(cf_pk(hj)));      // (hi.y hj.y, hi.x hj.y)
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(t) : "v"(p1), "v"(p2));                                     // (tx, ty)
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(cf_pk(P)), "v"(sc), "v"(t));  // P lam_inv - t
    return pk_cf(r);
}
#else
DS_HD cf herm_downdate_h(cf P, cf hi, cf hj, float lam_inv) { return herm_downdate_h_s(P, hi, hj, lam_inv); }
DS_HD cf herm_downdate(cf P, cf gi, cf gj, float lam_inv, float dls) { return herm_downdate_s(P, gi, gj, lam_inv, dls); }
DS_HD cf cfnmac_ca(cf acc, cf a, cf b) { return cfnmac_s(mk(acc.x, -acc.y), a, b); }
DS_HD cf cscale_c(cf a, float s) { return mk(a.x * s, -a.y * s); }
DS_HD cf cmul(cf a, cf b) { return cmul_s(a, b); }
DS_HD cf cmulc(cf a, cf b) { return cmulc_s(a, b); }
DS_HD cf cfma(cf acc, cf a, cf b) { return cfma_s(acc, a, b); }
DS_HD cf cfmac(cf acc, cf a, cf b) { return cfmac_s(acc, a, b); }
DS_HD cf cfnma(cf acc, cf a, cf b) { return cfnma_s(acc, a, b); }
DS_HD cf cfnmac(cf acc, cf a, cf b) { return cfnmac_s(acc, a, b); }
#endif
// Floor of a Cholesky pivot of A = R + delta I with R positive semi-definite: every pivot (a diagonal entry of a Schur complement of A)
// is >= lambda_min(A) >= delta in exact arithmetic, so the floor changes nothing there.  In fp32 a rank-deficient R (fewer distinct frames
// than microphones: a periodic input, the bench's replayed hops) leaves pivots of rounding-level size and either sign; floored at 1e-30
// they became r = 1e15 and the solve overflowed to NaN (found by bench.py --total-batch on a three-hop round, round 4).
DS_HD float pivot_floor(float delta) { return delta > 1e-30f ? delta : 1e-30f; }
// max(s, floor) as ONE v_max_f32 (written as a compare and a select against a run-time floor it compiled to six instructions per pivot:
// +8 % vector instructions in the MVDR frame kernel)
DS_HD float pivot_max(float s, float floor) { return __builtin_fmaxf(s, floor); }
DS_HD cf cconj(cf a) { return mk(a.x, -a.y); }
DS_HD cf cscale(cf a, float s) { return mk(a.x * s, a.y * s); }
DS_HD float cabs2(cf a) { return fma_(a.x, a.x, a.y * a.y); }
DS_HD cf cdiv(cf a, cf b) {
    const float d = 1.0f / cabs2(b);
    const cf n = cmulc(a, b);
    return mk(n.x * d, n.y * d);
}

// 1 - a for a smoothing constant given as a float: the complement is taken in double of the shortest decimal that rounds to `a`
// (0.9998f stands for 0.9998, whose complement is 2e-4; 1.0f - 0.9998f = 1.99974e-4 scales a covariance recursion by 1 - 1.3e-4).
// Host-side set-up helper (fill_params and the emulator's entry points).
inline float complement_of(float a) {
    const double d = (double)a;
    for (double scale = 1e1; scale <= 1e8; scale *= 10.0) {
        const double r = (double)(long long)(d * scale + 0.5) / scale;
        if ((float)r == a) return (float)(1.0 - r);
    }
    return (float)(1.0 - d);
}

DS_HD float fminf_(float a, float b) { return a < b ? a : b; }
DS_HD float fmaxf_(float a, float b) { return a > b ? a : b; }

enum { ALGO_FIXED = 0, ALGO_ADAPTIVE = 1, ALGO_GSC = 2, ALGO_AIC = 3, ALGO_ADAPTIVE_PF = 4 };   // ALGO_AIC: the SubbandGSC chain's tail (see aic_bin)
// ALGO_ADAPTIVE_PF: the adaptive beamformer's frame program followed, in the same per-bin phase, by the McMcra speech-presence gain on the
// same input frame (the post-filter convention of GSC.py:225,286): Y = w^H z * spp.G — "MVDR + post-filter" in one pass
DS_HD constexpr bool algo_has_mcra(int algo) { return algo == ALGO_ADAPTIVE || algo == ALGO_ADAPTIVE_PF; }
DS_HD constexpr bool algo_has_spp(int algo) { return algo == ALGO_GSC || algo == ALGO_ADAPTIVE_PF; }
enum { METHOD_SRC = 0, METHOD_DS = 1, METHOD_MVDR = 2, METHOD_TFGSC = 3 };

// Device-resident uniform counters of a chain stage, cnt = {frm_cnt, ell, first_frame, aux}: a later kernel of the same stream carries
// the advance in its arguments (thread 0 of block 0 applies it) instead of a launch of its own.  frames: frames to advance by (MCRA
// window L, mcra.py:52-56,72-74); aux moves by aux_add modulo aux_mod (FIR ping-pong parity, WPE delay-ring position).
struct TickArgs { int* cnt; int frames, L, aux_add, aux_mod; };
DS_HD void apply_tick(const TickArgs& t) {
    if (!t.cnt) return;
    int frm = t.cnt[0], ell = t.cnt[1];
    for (int i = 0; i < t.frames; ++i) {
        if (frm != 0 && ell % t.L == 0) ell = 0;
        frm += 1; ell += 1;
    }
    t.cnt[0] = frm; t.cnt[1] = ell;
    if (t.frames > 0) t.cnt[2] = 0;
    if (t.aux_mod > 0) t.cnt[3] = (t.cnt[3] + t.aux_add) % t.aux_mod;
}

// ---------------------------------------------------------------------------------------------
// Launch parameters (plain data; passed to the kernel by value)
// ---------------------------------------------------------------------------------------------
struct Params {
    const float* x;           // input samples (device)
    float* y;                 // output samples (device), [B][T*hop]
    long long x_batch_stride; // elements between utterances
    long long x_sample_stride;
    long long x_chan_stride;
    long long y_batch_stride;
    int T;                    // hops in this call
    int batch0;               // first utterance handled by block 0 (state / io index offset)
    vec4* bins;               // per-bin state [B][NF KP floats]: NF / 4 float4 planes [KP], then NF % 4 floats per bin as a narrow plane (StateLayout)
    float* tail_in;           // STFT overlap [B][M][hop]
    float* tail_out;          // OLA overlap  [B][hop]
    int* counters;            // [B][4]  {mcra frm_cnt, mcra ell, spp frm_cnt, reserved}
    const vec4* tables;       // Tables<NFFT> blob (twiddles, per-stage twiddles, sqrt-Hann window)
    const cf* steer;          // [K][M] steering vector a (adaptive/GSC) or weights W (fixed)
    long long steer_batch_stride;   // 0: one look direction shared by the batch
    int method;               // METHOD_* (adaptive) ; GSC: 0 = pass channel 0, else GSC
    int mcra_L;               // MCRA minimum-search window (mcra.py:25)
    float out_scale;          // hop / W0 (transform.py:479)
    float alpha_y, alpha_v;   // adaptivebeamformer.py:65-66
    float beta_y, beta_v;     // 1 - alpha as the reference's doubles give it (complement_of(): 1 - 0.9998f in fp32 is off by 1.3e-4 relative)
    float diag;               // adaptivebeamformer.py:89
    float diag_floor;         // pivot_floor(diag), formed by whoever fills diag: a kernel ARGUMENT (scalar register) — formed in the kernel it was a
                              // loop-invariant vector register held across the whole call (and the one value the one-pass MVDR + post-filter
                              // kernel spilled at four waves per SIMD)
    float gate;               // adaptivebeamformer.py:94
    unsigned gate_kinv;       // ~(number of leading bins whose gate may open), 0 = every bin: the `estPos` start-frame gate counts (frame, bin) slots
                              // (adaptivebeamformer.py:90-93: frameCount advances once per BIN), so the one frame in which the count runs out
                              // updates bins [0, r) only — gate_open(); the complement form keeps a zero-filled Params meaning 'no restriction'
    float mu;                 // GSC.py:202
    float* ref_pow;           // GSC, optional (null = off): [B][T][K][M] float, per frame and bin |Y|^2 of the canceller output in front of the
                              // post-filter gain and |U_i|^2 of the M - 1 blocking-matrix outputs — what GSC.py:281-283 hands to
                              // NsOmlsaMulti.estimation (DS_PARAM_REF_POWERS; utterance index = batch0 + block, as the state's)
    int rows;                 // single-channel transforms run one row per wavefront (StftRowsEngine / IstftRowsEngine): number of rows
    TickArgs tick;            // counters of an EARLIER stage of the chain to advance (stand-alone transform kernels only; cnt null = none)
    TickArgs tick2, tick3;    // two more (synthesis kernels only: the last launch of a DS_ALGO_WPE_MVDR group advances every counter of its step)
    // ALGO_AIC, the SubbandGSC chain's tail as one frame kernel: re-analysis of the M blocking-matrix outputs (x) -> multi-channel subband
    // NLMS canceller (SubbandLmsMc, 2 taps) -> synthesis (y).  The canceller's state stays where the DS_ALGO_SUBLMS operator keeps it.
    float* aic_st;            // [B][aic_NF][KP] planes: W (2 N M), tap buffer X (2 N M), smoothed input power P
    int aic_NF;
    const float* aic_d;       // desired-signal spectra, complex [B][T][K]; the canceller takes them one frame late (SubbandGSC.py:226) ...
    float* aic_dprev;         // ... complex [B][K]: the frame carried from the previous call
    const float* aic_p;       // update probability [B][T][K]
    int aic_pc, aic_norm;     // p -> 1 - p (SubbandGSC.py:232); normalised step (SubbandLmsMc.py:174-180)
    float aic_mu, aic_alpha, aic_reg;
    // ... with the synthesis of the blocking-matrix outputs in front of it: when aic_e is set the kernel takes the blocking filters' error
    // spectra instead of time samples (x unused), synthesises the M outputs itself (IstftEngine's arithmetic, overlap tails in aic_bmtail)
    // and hands them to the re-analysis through LDS; aic_bm (optional) receives the time-domain outputs
    const float* aic_e;       // complex [B * M][T][K]
    float* aic_bmtail;        // [B][M][hop]
    float* aic_bm;            // [B][M][T * hop] or null
    // StftEngine<.., CDR = true>, the analysis of the SubbandGSC chain's front end with McCDR (mccdr_frame) as its per-bin program: the
    // workgroup holds the utterance, so the MCRA stencil comes from LDS and McSpp's band average of 1 - Gamma is a sum over LDS
    float* cdr_st;            // the DS_ALGO_MCSPP stage's planes [B][cdr_NF][KP]; rows 0..8 are McCDR's (p1, p2, x12, MCRA S..lambda_d)
    int cdr_NF;
    int cdr_frm, cdr_ell, cdr_L;   // the stage's uniform counters before this call (host mirror: the SubbandGSC chain's launch is never part of a replayed graph)
    const int* cdr_cnt;            // ... or, non-null, their device copy {frm, ell, ...} (ds_handle::dev_cnt of the McSpp stage): the notebook-MVDR chain replays as a hipGraph
    const float* cdr_fn;      // diffuse coherence of the microphone pair, [K]
    // StftEngine<.., CDR, 2, FRONT = true>: the chain's WHOLE front end in that kernel — FilterDcNotch16 per channel (feature.py:32-49), the
    // TimeAlignment FIR bank and the channel mean (fixedbeamformer.py:13-93, SubbandGSC.py:143) run on the hop in LDS, in front of the
    // analysis: x is then the RAW input (strided [B][M][n]).  Same arithmetic, same order as ds_dcnotch_kernel / ds_fir_kernel
    const float* fe_coef;     // FIR taps [L][M]
    int fe_L;
    double* fe_mem;           // notch memories [B][M][2] (doubles: ds_ops.hpp td_dcnotch)
    const float* fe_cache_in; // FIR history [B][M][L-1] before this call ...
    float* fe_cache_out;      // ... and after it (the other half of the ping-pong pair)
    float* fe_fixed;          // channel mean of the aligned channels, [B][T * hop] (the fixed beamformer's block)
    double fe_radius;         // the decimal the caller wrote, as a double (decimal_double)
    float* cdr_gamma;         // out: Gamma [B][T][K]
    float* cdr_qavg;          // out: mean of 1 - Gamma over the 500-2000 Hz band, [B][T]  (mcspp.py:258-260)
};

// number of per-bin state floats / planes
// lanes per plane row of the per-bin state arrays ([..][plane][KP] of float4 or float): K bins rounded up to 8, so that a row of 16-byte
// words is a whole number of 128-byte lines (264 x 16 B = 33 lines at 257 bins; the 260 of rounds 1 - 3 were 32.5: every other row
// started in the middle of a line, and a wavefront's 1 KB of a row was 7 whole lines and two halves.  With the frame kernels'
// non-temporal state traffic the aligned rows are +4.5 % at 1024 utterances, +3 % at 4096; profiles/r04a/plane_rows_ab.txt)
DS_HD constexpr int plane_len(int K) { return (K + 7) & ~7; }

template <int M, int ALGO, bool RYY> struct StateLayout {
    static constexpr int NF =
        ALGO == ALGO_ADAPTIVE ? (M * M + 5 + (RYY ? M * M : 0))
        : ALGO == ALGO_ADAPTIVE_PF ? (M * M + 5 + M * (M + 1))     // the adaptive program's floats, then McMcra's Phi_yy, Phi_vv (packed symmetric)
        : ALGO == ALGO_GSC    ? (M * (M + 1) + 2 * (M - 1))
        : ALGO == ALGO_AIC    ? (8 * M + 1)            // 2-tap M-channel canceller: W, X (2 M complex each), P
                              : 0;
    static constexpr int NP = (NF + 3) / 4;
    // in memory: NPF full 16-byte planes [NPF][KP] of float4, then the NF % 4 floats that are left as ONE narrow plane [KP][RT] — not a
    // sixth / seventh float4 plane with spare words: the adaptive kernel's 21 floats per bin are 84 bytes each way instead of 96
    // (one hop per call moves the whole state, and the launch is bound by exactly those bytes); ust(KP) floats per utterance
    static constexpr int NPF = NF / 4, RT = NF % 4;
    // floats between utterances: NF KP rounded up to a 128-byte line, so that every utterance's planes start on a line like the first one's
    static constexpr long long ust(int KP) { return ((long long)NF * KP + 31) & ~31LL; }
    // ADAPTIVE float map
    static constexpr int R_DIAG = 0;                 // M reals
    static constexpr int R_OFF = M;                  // M(M-1)/2 complex, (i<j) row-major
    static constexpr int MC_S = M * M;               // S, Smin, Stmp, p, lambda_d
    static constexpr int RYY_DIAG = M * M + 5;
    static constexpr int RYY_OFF = M * M + 5 + M;
    // GSC float map
    static constexpr int PYY = 0;                    // sym packed (i<=j) row-major, M(M+1)/2
    static constexpr int PVV = M * (M + 1) / 2;
    static constexpr int GA = M * (M + 1);           // (M-1) complex
    // ADAPTIVE_PF float map: [0, M M + 5) as ADAPTIVE, then
    static constexpr int PF_PYY = M * M + 5;
    static constexpr int PF_PVV = M * M + 5 + M * (M + 1) / 2;
};

DS_HD constexpr int off_index(int i, int j, int M) {   // i<j -> index among strictly-upper entries
    return i * M - (i * (i + 1)) / 2 + (j - i - 1);
}
DS_HD constexpr int sym_index(int i, int j, int M) {   // i<=j -> index among upper-incl-diag entries
    return i * M - (i * (i - 1)) / 2 + (j - i);
}

// Constant tables, laid out identically in HBM (built once on the host, ds_tables.hpp) and in LDS so the
// prologue copies them with 16-byte loads/stores.
template <int NFFT> struct Tables {
    static constexpr int N = NFFT, NC = NFFT / 2;
    static constexpr int NSTW = NC == 512 ? 512 : 128;   // entries used: [4, 2 * (largest Ns))
    cf tw[NC + 2];        // exp(-2 pi j i / N), i = 0..NC (+1 pad): split / merge of the packed real transform
    vec4 stw[NSTW];       // per-stage twiddles, contiguous per stage: stw[Ns + k] = (w1, w2), w_r = exp(-2 pi j k r / (Ns R))
    float win[N];         // sqrt-Hann
    static constexpr int NV4 = ((NC + 2) * 8 + NSTW * 16 + N * 4) / 16;
};

template <int NFFT, int M, int NYQF = 4> struct Shared {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2;
    alignas(16) float xbuf[M][N];     // two halves: [old hop | new hop], roles swap every frame
    static constexpr int NCP = NC + NC / 4;   // room for the bank-conflict padding of the early FFT stages
    cf fa[M][NCP];
    cf fb[M][NCP];
    alignas(16) Tables<NFFT> tb;
    float pw[K + 3];      // |Z_0|^2 for the MCRA frequency stencil
    cf Y[K + 1];          // beamformer output spectrum
    alignas(16) float tail[HOP];      // overlap-add tail
    alignas(16) float nyq[NYQF];      // per-bin state of the Nyquist bin (k = N/2)
    float zn[M];                      // ... and its input Z[N/2][m] (real)
};

template <int M, int ALGO, bool RYY, int NPRE> struct Regs {
    cf Z[M];
    float st[StateLayout<M, ALGO, RYY>::NP * 4 + 1];
    vec4 pre[NPRE];
    const vec4* xp[NPRE];     // this lane's read position in the input stream (advanced one hop per frame)
    vec4 nyq;                 // prologue: plane `tid` of the Nyquist bin on its way to LDS
    cf ad, adn;               // ALGO_AIC: desired-signal sample of this frame for this lane's bin / for the Nyquist bin (lane NYQ_TID)
    float apk, apkn;          // ... and the update probability
    float bmt[2 * M];         // ... and this lane's two samples of the M blocking-matrix overlap tails (lanes < NC / 2; aic_e mode)
    float cdr[9], cdrn[9];    // StftEngine<.., CDR>: McCDR's state of this lane's bin / of the Nyquist bin (lane 0)
    double nm0, nm1;          // StftEngine<.., FRONT>: the DC notch memory of channel `tid` (lanes < M)
};

// state-plane accessors.  Every state line is read once and written once per launch and is next touched by the following launch,
// possibly from another XCD (whose L2 is private): non-temporal on the device, so the lines stream through instead of sitting
// dirty in L2 until the end-of-kernel write-back (measured: +4 % at B = 1024 and +11 % at B = 4096, one hop per call).
// DS_PLAIN_STATE restores ordinary loads / stores for A/B runs.
DS_HD void store_state(vec4* dst, const vec4& v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_PLAIN_STATE) && !defined(DS_PLAIN_STATE_STORE)
    typedef float f4_t __attribute__((ext_vector_type(4)));
    f4_t q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    __builtin_nontemporal_store(q, reinterpret_cast<f4_t*>(dst));
#else
    *dst = v;
#endif
}
DS_HD vec4 load_state(const vec4* src) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_PLAIN_STATE) && !defined(DS_PLAIN_STATE_LOAD)
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const f4_t q = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(src));
    vec4 v; v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
    return v;
#else
    return *src;
#endif
}
// the same for one complex word (rows of 16 of them: one line per instruction in the RLS-WPE blocks' line layout, ds_wpe.hpp)
DS_HD void store_state(cf* dst, const cf& v) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_PLAIN_STATE) && !defined(DS_PLAIN_STATE_STORE)
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t q; q.x = v.x; q.y = v.y;
    __builtin_nontemporal_store(q, reinterpret_cast<f2_t*>(dst));
#else
    *dst = v;
#endif
}
DS_HD cf load_state(const cf* src) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_PLAIN_STATE) && !defined(DS_PLAIN_STATE_LOAD)
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t q = __builtin_nontemporal_load(reinterpret_cast<const f2_t*>(src));
    cf v; v.x = q.x; v.y = q.y;
    return v;
#else
    return *src;
#endif
}

// ---------------------------------------------------------------------------------------------
// FFT stages (Stockham autosort, radix 4 / radix 2), all channels of the block at once
// ---------------------------------------------------------------------------------------------
// a + j d and a - j d: the quarter turn is a half swap with one negated half — one packed add each on the device (the compiler forms
// both full sums and then picks halves with v_mov); the same additions either way
template <int SIGN> DS_HD cf cadd_jd(cf a, cf d) {        // a + SIGN * j * d
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_SCALAR_COMPLEX)
    cf2_t r;
    if constexpr (SIGN > 0) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(d)));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(d)));
    return pk_cf(r);
#else
    return SIGN > 0 ? mk(a.x - d.y, a.y + d.x) : mk(a.x + d.y, a.y - d.x);
#endif
}

// a + conj(b), a - conj(b), d / (2j): the merge / split of the packed real transform; one packed instruction each on the device
DS_HD cf cadd_c(cf a, cf b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_SCALAR_COMPLEX)
    cf2_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)));
    return pk_cf(r);
#else
    return mk(a.x + b.x, a.y + (-b.y));
#endif
}
DS_HD cf csub_c(cf a, cf b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_SCALAR_COMPLEX)
    cf2_t r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(cf_pk(a)), "v"(cf_pk(b)));
    return pk_cf(r);
#else
    return mk(a.x - b.x, a.y - (-b.y));
#endif
}
DS_HD cf cdiv_2j(cf d) {                         // (0.5 d.y, -0.5 d.x)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_SCALAR_COMPLEX)
    cf2_t r;
    const cf2_t h = {0.5f, -0.5f};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(cf_pk(d)), "v"(h));
    return pk_cf(r);
#else
    return mk(0.5f * d.y, -0.5f * d.x);
#endif
}

template <int R, int SIGN> DS_HD void butterfly(cf* v) {
    if constexpr (R == 4) {
        cf t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]), t2 = cadd(v[1], v[3]);
        cf d = csub(v[1], v[3]);
        // t3 = (-/+ j) d;  v1 = t1 + t3, v3 = t1 - t3
        v[0] = cadd(t0, t2); v[1] = cadd_jd<(SIGN < 0 ? -1 : +1)>(t1, d); v[2] = csub(t0, t2); v[3] = cadd_jd<(SIGN < 0 ? +1 : -1)>(t1, d);
    } else {
        cf a = v[0], b = v[1];
        v[0] = cadd(a, b); v[1] = csub(a, b);
    }
}

// LDS index padding of an FFT buffer.  The radix-4 Stockham stage with sub-transform size Ns writes
// out[(j - k) * 4 + k + r * Ns]: for Ns = 1 a stride of 4 complex, for Ns = 4 runs of 4 complex every 16 —
// 4-way bank conflicts for ds_write_b64 on a linear buffer.  PAD 1 (i + i/16) makes the Ns = 1 pattern
// conflict-free, PAD 2 (i + 4*(i/16)) the Ns = 4 pattern; later stages write contiguous runs (PAD 0).
template <int PAD> DS_HD int padi(int i) {
    if constexpr (PAD == 1) return i + (i >> 4);
    else if constexpr (PAD == 2) return i + ((i >> 4) << 2);
    else return i;
}

// in/out: [MCH][NCP].  FROM = 1: read windowed real samples packed as (x[2n], x[2n+1]) from xbuf.  FROM = 2 (inverse, one channel): form
// point n of the packed spectrum from the output bins sh.Y[n], sh.Y[NC - n] on the fly (E + j O of the real-transform merge, with
// the Nyquist bin taken as zero: see Engine::run).
// offset of entry i + d from entry i in a padded buffer, for the displacements a stage uses: a multiple of 16 moves the padding term by a
// constant; a displacement that stays inside i's own run of 16 leaves it alone (the caller guarantees which case holds)
template <int PAD> DS_HD constexpr int pad_step16(int d) { return PAD == 1 ? d + (d >> 4) : PAD == 2 ? d + ((d >> 4) << 2) : d; }

// One butterfly job of a stage = (channel ch, butterfly j).  fft_load brings its R inputs and the stage's twiddle pair into registers;
// fft_finish applies the twiddles and the butterfly and writes the R outputs.  fft_stage runs the two back to back for every job of a
// thread; the hop-pipelined engine (ds_pipe.hpp) runs other work between them and transforms in place (the whole channel is in its
// wave's registers between the two halves).
template <int NFFT, int R, int SIGN, int FROM, int PIN, class ShT>
DS_HD void fft_load(ShT& sh, const cf* in, int ch, int j, int Ns, int old_half, cf* v, vec4& wv) {
    constexpr bool FROM_X = FROM == 1;
    constexpr int NC = NFFT / 2, NB = NC / R, HOP = NFFT / 2, NCP = ShT::NCP;
    static_assert(NB % 16 == 0, "the inputs of a butterfly are NB apart: a constant step in a padded buffer");
    constexpr int RSTEP = pad_step16<PIN>(NB);
    // the R inputs of butterfly j sit at n = j + r NB: one address, constant steps
    if constexpr (FROM == 3) {
        // hop = NFFT / 4 (Transform with 75 % overlap): xbuf is a ring of four quarter-frames and `old_half` the slot of the oldest;
        // the r-th input of a radix-4 butterfly of the first stage lies in the r-th quarter of the frame
        static_assert(R == 4 && 2 * NB == NFFT / 4, "first stage of the quarter-hop transform");
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = 2 * j + r * 2 * NB, pos = ((old_half + r) & 3) * (NFFT / 4) + 2 * j;
            v[r] = mk(sh.tb.win[s] * sh.xbuf[ch][pos], sh.tb.win[s + 1] * sh.xbuf[ch][pos + 1]);
        }
    } else if constexpr (FROM_X) {
        static_assert(R == 4 || R == 2, "radix");
        const int s0 = 2 * j;                                               // s = s0 + r * 2 NB; 2 NB = HOP / 2 (R = 4) or HOP (R = 2)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = s0 + r * 2 * NB;
            const bool first = r * 2 * NB + 2 * NB <= HOP;                  // compile-time: the whole range of s for this r is below HOP
            const int pos = first ? old_half * HOP + s : (old_half ^ 1) * HOP + (s - HOP);
            v[r] = mk(sh.tb.win[s] * sh.xbuf[ch][pos], sh.tb.win[s + 1] * sh.xbuf[ch][pos + 1]);
        }
    } else if constexpr (FROM == 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = j + r * NB;
            const cf* Yb = in;                                          // the output spectrum Y[0 .. NC]
            const cf A = Yb[n];
            cf S, Dm;                                               // A + conj(B), A - conj(B), B = Y[NC - n]
            if (NC == 256 && r == 0) {                              // NC == 256: the Nyquist bin (n = 0's partner) is added later: B = 0 there
                const cf Bc = n == 0 ? mk(0.0f, 0.0f) : cconj(Yb[NC - n]);
                S = cadd(A, Bc); Dm = csub(A, Bc);
            } else {
                const cf B = Yb[NC - n];
                S = cadd_c(A, B); Dm = csub_c(A, B);
            }
            const cf E = cscale(S, 0.5f);
            const cf O = cmul(cscale(Dm, 0.5f), cconj(sh.tb.tw[n]));
            v[r] = cadd_jd<+1>(E, O);                               // E + j O
        }
    } else {
        const cf* rp = in + ch * NCP + padi<PIN>(j);
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = rp[r * RSTEP];
    }
    if (Ns > 1) wv = sh.tb.stw[Ns + (j & (Ns - 1))];        // consecutive lanes -> consecutive k: conflict-free
}

template <int NFFT, int R, int SIGN, int POUT, class ShT>
DS_HD void fft_finish(ShT&, cf* out, int ch, int j, int Ns, cf* v, const vec4& wv) {
    constexpr int NCP = ShT::NCP;
    const int k = j & (Ns - 1);
    if (Ns > 1) {
        cf w1 = mk(wv.x, wv.y);
        if (SIGN > 0) w1 = cconj(w1);
        if constexpr (R == 4) {
            cf w2 = mk(wv.z, wv.w);
            if (SIGN > 0) w2 = cconj(w2);
            cf w3 = cmul(w1, w2);
            v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3);
        } else {
            v[1] = cmul(v[1], w1);
        }
    }
    butterfly<R, SIGN>(v);
    // the R outputs sit at j0 + r Ns, j0 = (j - k) R + k: Ns a multiple of 16 is a constant step in a padded buffer; for Ns R <= 16 the
    // R outputs share j0's run of 16 (j0 - k is a multiple of Ns R, which divides 16), so the padding term does not move
    const int j0 = (j - k) * R + k;
    cf* wp = out + ch * NCP + padi<POUT>(j0);
    if (POUT == 0 || (Ns & 15) == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) wp[pad_step16<POUT>(r * Ns)] = v[r];
    } else if (Ns * R <= 16) {
#pragma unroll
        for (int r = 0; r < R; ++r) wp[r * Ns] = v[r];
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) out[ch * NCP + padi<POUT>(j0 + r * Ns)] = v[r];
    }
}

template <int NFFT, int M, int R, int SIGN, int FROM, int PIN, int POUT, class ShT>
DS_HD void fft_stage(int tid, int nt, ShT& sh, const cf* in, cf* out, int Ns, int old_half, int MCH) {
    constexpr int NC = NFFT / 2, NB = NC / R;
    DS_ASSUME(tid >= 0 && tid < nt);                                     // a one-trip loop where MCH * NB <= nt: no loop-carried addresses
    for (int idx = tid; idx < MCH * NB; idx += nt) {
        // (unsigned: the split into channel and butterfly is a shift and a mask, and j's range is known to the compiler)
        const int ch = (int)((unsigned)idx / (unsigned)NB), j = (int)((unsigned)idx % (unsigned)NB);
        cf v[R];
        vec4 wv;
        wv.x = wv.y = wv.z = wv.w = 0.0f;
        const cf* src = in;
        if constexpr (FROM == 2) src = reinterpret_cast<const cf*>(&sh.Y[0]);       // the inverse merge reads the output spectrum
        fft_load<NFFT, R, SIGN, FROM, PIN>(sh, src, ch, j, Ns, old_half, v, wv);
        fft_finish<NFFT, R, SIGN, POUT>(sh, out, ch, j, Ns, v, wv);
    }
}

// ---------------------------------------------------------------------------------------------
// MCRA (one bin) — mcra.py:27-77.  k: bin, K: half_bin.  st: S,Smin,Stmp,p,lambda_d.
// frm_cnt / ell are the frame-level counters *before* this frame; `reset` = (ell % L == 0).
// ---------------------------------------------------------------------------------------------
DS_HD void mcra_bin(float* st, int k, int K, float Ykm1, float Yk, float Ykp1, int frm_cnt, bool reset, int L) {
    const float alpha_s = 0.8f, one_m_alpha_s = (float)(1.0 - 0.8), delta_s = 5.0f;
    const float alpha_p = 0.2f, one_m_alpha_p = (float)(1.0 - 0.2), alpha_d = 0.95f, one_m_alpha_d = (float)(1.0 - 0.95);
    const float p_max = 0.999f, p_min = 1e-3f;
    const float S0 = st[0], Smin0 = st[1], Stmp0 = st[2], p0 = st[3], lam0 = st[4];
    // the recursion of an interior bin, evaluated by every lane and selected below: as branches the three cases cost more in copies of
    // the state between the paths than the few lanes that take the short ones save
    const float Sf = fma_(Ykp1, 0.25f, fma_(Yk, 0.5f, Ykm1 * 0.25f));               // :46
    const float S1 = fma_(alpha_s, S0, one_m_alpha_s * Sf);                          // :47
    const float Smin_a = fminf_(Smin0, S1), Stmp_a = fminf_(Stmp0, S1);             // :49-50
    const float Smin1 = reset ? fminf_(Stmp_a, S1) : Smin_a, Stmp1 = reset ? S1 : Stmp_a;   // :52-56
    // Sr = S / (Smin + 1e-6) > delta (:58-63) as S > delta (Smin + 1e-6): the quotient is used for nothing else, and an IEEE fp32 division is ten
    // instructions of this 200-instruction program (both sides positive; at an exact tie the two forms may differ in the last ulp — as the fp32 S
    // already differs from the reference's double)
    const float I = S1 > delta_s * (Smin1 + 1e-6f) ? 1.0f : 0.0f;
    float p1 = fma_(alpha_p, p0, one_m_alpha_p * I);                                // :65-67
    if (frm_cnt < 2 * L) p1 = 0.0f;                                                 // :68-69
    const bool first = frm_cnt == 0, init = first && k < K - 1;                     // :38-41,68-69
    const bool upd = !first && k > 0 && k < K - 1;
    const float S = upd ? S1 : S0;
    const float Smin = upd ? Smin1 : init ? Yk : Smin0;
    const float Stmp = upd ? Stmp1 : init ? Yk : Stmp0;
    float lam = init ? Yk : lam0;
    float p = upd ? p1 : (init || (!first && k == 0)) ? 0.0f : p0;                  // :43-45
    p = fmaxf_(fminf_(p, p_max), p_min);                                              // :70
    if (k == K - 1) lam = 1e-8f;                                                      // :73
    const float at = fma_(one_m_alpha_d, p, alpha_d);                                     // Base :57
    lam = fma_(at, lam, (1.0f - at) * Yk);                                                // Base :60
    st[0] = S; st[1] = Smin; st[2] = Stmp; st[3] = p; st[4] = lam;
}

// McCDR for one bin and frame (noise_estimation/mccdr.py:120-175 with BinauralEnhancement.py:28-61): recursive auto / cross spectra of
// microphones 1 and 2, the coherent-to-diffuse ratio against the diffuse coherence Fn, and the MCRA speech presence probability of
// microphone 0 (stencil ym, y0, yp = |Y0|^2 at bins k - 1, k, k + 1); returns Gamma = sqrt(CDR^2 p).  State: p1, p2, x12, mc[5].
DS_HD float mccdr_frame(float& p1, float& p2, cf& x12, float* mc, cf y1, cf y2, float Fn, float ym, float y0, float yp, int k, int K,
                        int frm, bool reset, int L) {
    p1 = fma_(0.9f, p1, (float)(1.0 - 0.9) * cabs2(y1));                           // BinauralEnhancement.py:49-52
    p2 = fma_(0.9f, p2, (float)(1.0 - 0.9) * cabs2(y2));
    const cf c12 = cmulc(y1, y2);
    x12 = mk(fma_(0.9f, x12.x, (float)(1.0 - 0.9) * c12.x), fma_(0.9f, x12.y, (float)(1.0 - 0.9) * c12.y));   // :55-61
    // coherent-to-diffuse ratio in double: with |Fx| -> 1 and Fn -> 1 (the lowest bins of a real recording) the radicand is the
    // difference of nearly equal terms
    const double rn = 1.0 / sqrt((double)p1 * (double)p2);
    const double Fxr = (double)x12.x * rn, Fxi = (double)x12.y * rn;               // updateMSC :28
    const double Fx2 = __builtin_fma(Fxr, Fxr, Fxi * Fxi);
    const double Fnd = (double)Fn, Fn2d = Fnd * Fnd;
    const double rad = Fn2d * (Fxr * Fxr) - Fn2d * Fx2 + Fn2d - 2.0 * Fnd * Fxr + Fx2;
    const double den = Fx2 - 1.0 < -1e-3 ? Fx2 - 1.0 : -1e-3;
    double Gd = (Fnd * Fxr - Fx2 - sqrt(rad)) / den;                               // mccdr.py:141-145
    Gd = Gd * Gd;
    if (Gd > 1.0) Gd = 1.0;                                                        // :160-161 (NaN stays NaN like numpy)
    if (Gd < 0.0) Gd = 1e-3;
    mcra_bin(mc, k, K, ym, y0, yp, frm, reset, L);                                 // :174 (L = 65)
    return (float)sqrt(Gd * (double)mc[3]);                                        // :175
}

// Hermitian packed (diag reals d[M], strictly-upper complex o[]) helpers ------------------------
template <int M> DS_HD cf herm_get(const float* d, const float* o, int i, int j) {   // element (i,j)
    if (i == j) return mk(d[i], 0.0f);
    if (i < j) { int q = off_index(i, j, M); return mk(o[2 * q], o[2 * q + 1]); }
    int q = off_index(j, i, M); return mk(o[2 * q], -o[2 * q + 1]);
}

// rank-1 recursive update  R <- a R + b z z^H   (adaptivebeamformer.py:86-88,97-99)
template <int M> DS_HD void herm_rank1(float* d, float* o, const cf* z, float a, float b) {
#pragma unroll
    for (int i = 0; i < M; ++i) d[i] = fma_(a, d[i], b * cabs2(z[i]));
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i + 1; j < M; ++j) {
            const int q = off_index(i, j, M);
            const cf zz = cmulc(z[i], z[j]);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DS_SCALAR_COMPLEX) && !defined(DS_RANK1_SCALAR)
          if constexpr (M >= 5) {      // (up to four microphones the vectoriser's own packing has no moves to lose, and the Nyquist bin's LDS-resident state keeps its paired reads)
            // the same two roundings per word as the scalar statements below, as ONE packed multiply and ONE packed multiply-add on the (re, im)
            // pair the product arrives in: left to the vectoriser it paired words of NEIGHBOURING elements and moved them together first (two
            // v_mov per element: 60 of the 6-microphone McSpp operator's 1080 vector instructions per frame)
            cf2_t t, r;
            // (the two words read one by one and pinned: read as a pair, the off-diagonal words that start at an odd float of the state became
            // 8-byte accesses across the 16-byte plane groups and the planes stayed in private memory — the adaptive_bin note on TFGSC)
            float ox = o[2 * q], oy = o[2 * q + 1];
            asm volatile("" : "+v"(ox), "+v"(oy));
            const cf2_t ab = {a, b}, ov = {ox, oy};
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(t) : "v"(cf_pk(zz)), "v"(ab));                       // (zz.x b, zz.y b)
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(ov), "v"(ab), "v"(t));               // (o.x a + t.x, o.y a + t.y)
            o[2 * q] = r.x; o[2 * q + 1] = r.y;
          } else {
            o[2 * q] = fma_(a, o[2 * q], b * zz.x);
            o[2 * q + 1] = fma_(a, o[2 * q + 1], b * zz.y);
          }
#else
            o[2 * q] = fma_(a, o[2 * q], b * zz.x);
            o[2 * q + 1] = fma_(a, o[2 * q + 1], b * zz.y);
#endif
        }
}

// Cholesky factor of A = R + diag*I (Hermitian PD), L lower: Ld real diagonal (stored inverted),
// Lo strictly-lower complex packed by (i>j) -> off_index(j,i).
template <int M> struct Chol {
    float inv[M];
    cf lo[M * (M - 1) / 2 + 1];
    DS_HD cf L(int i, int j) const { return lo[off_index(j, i, M)]; }   // i>j
    DS_HD void factor(const float* d, const float* o, float diag) {
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float s = d[j] + diag;
#pragma unroll
            for (int k = 0; k < j; ++k) { const cf l = L(j, k); s = fma_(-l.x, l.x, fma_(-l.y, l.y, s)); }
            s = pivot_max(s, pivot_floor(diag));     // a pivot of R + diag I (R >= 0) is >= diag in exact arithmetic: see pivot_floor()
#if defined(__HIP_DEVICE_COMPILE__)
            const float r = __builtin_amdgcn_rsqf(s);      // (s >= 1e-30: see mvdr_output)
#else
            const float r = 1.0f / sqrtf(s);
#endif
            inv[j] = r;
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                // A_ij = conj(A_ji), the stored word: conjugated inside the first multiply-add (or the scale) instead of in a register pair
                const int w = off_index(j, i, M);
                const cf aji = mk(o[2 * w], o[2 * w + 1]);
                if (j == 0) {
                    lo[w] = cscale_c(aji, r);
                } else {
                    cf a = cfnmac_ca(aji, L(i, 0), L(j, 0));
#pragma unroll
                    for (int k = 1; k < j; ++k) a = cfnmac(a, L(i, k), L(j, k));
                    lo[w] = cscale(a, r);
                }
            }
        }
    }
    // solve A v = b
    DS_HD void solve(const cf* b, cf* v) const {
        cf u[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cf a = b[i];
#pragma unroll
            for (int k = 0; k < i; ++k) a = cfnma(a, L(i, k), u[k]);
            u[i] = cscale(a, inv[i]);
        }
#pragma unroll
        for (int i = M - 1; i >= 0; --i) {
            cf a = u[i];
#pragma unroll
            for (int k = i + 1; k < M; ++k) a = cfnmac(a, v[k], L(k, i));
            v[i] = cscale(a, inv[i]);
        }
    }
    // L -> L^-1 in place (lower triangular, real diagonal inv[]): column by column, entry (i, j) from row i of L to the right of column j
    // (not overwritten yet) and the finished part of column j.  After this L(i, j) reads L^-1 and solve() must not be used; apply() and
    // trace_with() take its place — one triangular inverse (M (M^2 - 1) / 6 complex multiply-adds) instead of a substitution pair per
    // right-hand side, and nothing but the factor itself stays live between the right-hand sides
    DS_HD void invert() {
#pragma unroll
        for (int j = 0; j < M; ++j)
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                cf a = cscale(L(i, j), inv[j]);
#pragma unroll
                for (int k = j + 1; k < i; ++k) a = cfma(a, L(i, k), L(k, j));
                lo[off_index(j, i, M)] = cscale(a, -inv[i]);
            }
    }
    // after invert(): v = A^-1 b = L^-H (L^-1 b)
    DS_HD void apply(const cf* b, cf* v) const {
        cf u[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cf a = cscale(b[i], inv[i]);
#pragma unroll
            for (int j = 0; j < i; ++j) a = cfma(a, L(i, j), b[j]);
            u[i] = a;
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
            cf a = cscale(u[j], inv[j]);
#pragma unroll
            for (int i = j + 1; i < M; ++i) a = cfmac(a, u[i], L(i, j));      // a + conj(L_ij) u_i: the same products in the same nesting as cfma(a, conj(L), u)
            v[j] = a;
        }
    }
    // after invert(): Re tr(A^-1 R) for a Hermitian-packed R = sum over the rows r_i of L^-1 of the quadratic forms r_i R r_i^H
    DS_HD float trace_with(const float* d, const float* o) const {
        float tr = 0.0f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            // row by row (the products of all rows at once are ~100 live registers): row i's entries go through a pin behind row i - 1's sum
            if (i >= 2) {
                DS_PIN(tr);
#pragma unroll
                for (int j = 0; j < i; ++j) { cf& l = const_cast<cf&>(lo[off_index(j, i, M)]); DS_PIN2(l.x, l.y); }
            }
            float q = inv[i] * inv[i] * d[i], c2 = 0.0f;
#pragma unroll
            for (int j = 0; j < i; ++j) {
                const cf rj = L(i, j);
                q = fma_(cabs2(rj), d[j], q);
                { const int w = off_index(j, i, M); c2 = fma_(rj.x * inv[i], o[2 * w], fma_(-(rj.y * inv[i]), o[2 * w + 1], c2)); }   // Re(r_ij R_ji... ) with r_ii real
#pragma unroll
                for (int k = j + 1; k < i; ++k) {
                    const cf c = cmulc(rj, L(i, k));                       // r_ij conj(r_ik)
                    const int w = off_index(j, k, M);
                    c2 = fma_(c.x, o[2 * w], fma_(-c.y, o[2 * w + 1], c2));   // Re(r_ij R_jk conj(r_ik))
                }
            }
            tr += fma_(2.0f, c2, q);
        }
        return tr;
    }
};

// MVDR output without forming the inverse or the weights:
//   Y = sum_m conj(w_m) z_m,  w = A^-1 a / (a^H A^-1 a),  A = R + diag*I = L L^H
//     = (u^H t) / (u^H u),    u = L^-1 a,  t = L^-1 z
// (adaptivebeamformer.py:103-112,119-120 + beamformer.py:325-326).  Right-looking Cholesky on a working
// copy of A with both forward substitutions fused into the column sweep: no back-substitution,
// no stored factor, real denominator.
#ifdef DS_SOLVE_FP64
// Measured variant (scratch/build_variant.sh WORK fp64 "-DDS_SOLVE_FP64"; docs/DESIGN_rounds_1_2.md section 3): the same fused sweep in double on the
// fp32 state, the reference's complex128 arithmetic (adaptivebeamformer.py:103-104).  Not the default: the fp32 sweep is already inside
// the 1e-4 RMS bar and this one costs registers (occupancy) and the fp64 vector rate.
template <int M> DS_HD cf mvdr_output(const float* d, const float* o, float diag, const cf* a, const cf* z) {
    struct dc { double x, y; };
    auto F = [](double p, double q, double r) { return __builtin_fma(p, q, r); };
    double Ad[M];
    dc Al[M * (M - 1) / 2 + 1], u[M], t[M];
#pragma unroll
    for (int i = 0; i < M; ++i) { Ad[i] = (double)d[i] + (double)diag; u[i] = dc{a[i].x, a[i].y}; t[i] = dc{z[i].x, z[i].y}; }
#pragma unroll
    for (int q = 0; q < M * (M - 1) / 2; ++q) Al[q] = dc{o[2 * q], -(double)o[2 * q + 1]};
    double nu = 0.0, utx = 0.0, uty = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const double sj = Ad[j] > 1e-30 ? Ad[j] : 1e-30;
        const double r = 1.0 / __builtin_sqrt(sj);
        const dc uj = dc{u[j].x * r, u[j].y * r}, tj = dc{t[j].x * r, t[j].y * r};
        nu = F(uj.x, uj.x, F(uj.y, uj.y, nu));
        utx = F(tj.x, uj.x, F(tj.y, uj.y, utx));                 // += conj(u_j) t_j
        uty = F(tj.y, uj.x, F(-tj.x, uj.y, uty));
        dc Lc[M];
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            const dc A = Al[off_index(j, i, M)];
            Lc[i] = dc{A.x * r, A.y * r};
            u[i] = dc{F(-Lc[i].x, uj.x, F(Lc[i].y, uj.y, u[i].x)), F(-Lc[i].x, uj.y, F(-Lc[i].y, uj.x, u[i].y))};
            t[i] = dc{F(-Lc[i].x, tj.x, F(Lc[i].y, tj.y, t[i].x)), F(-Lc[i].x, tj.y, F(-Lc[i].y, tj.x, t[i].y))};
        }
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            Ad[i] = F(-Lc[i].x, Lc[i].x, F(-Lc[i].y, Lc[i].y, Ad[i]));
#pragma unroll
            for (int k = j + 1; k < i; ++k) {
                const int q = off_index(k, i, M);                // A_ik -= L_ij conj(L_kj)
                Al[q] = dc{F(-Lc[i].x, Lc[k].x, F(-Lc[i].y, Lc[k].y, Al[q].x)), F(-Lc[i].y, Lc[k].x, F(Lc[i].x, Lc[k].y, Al[q].y))};
            }
        }
    }
    const double inv = 1.0 / nu;
    return mk((float)(utx * inv), (float)(uty * inv));
}
#else
// The same sweep as a resumable object (init, one column at a time, finish) for the hop-pipelined engine (ds_pipe.hpp), which runs the
// columns between the stages of the neighbouring hops' transforms.  The operations and their order are mvdr_output()'s, word for word
// (bit-identical results: tests/test_kernel_emul.py::test_emul_pipelined_engine_equals_the_frame_engine and its GPU twin).
template <int M> struct MvdrSweep {
    float dg;
    float Ad[M];
    cf Al[M * (M - 1) / 2 + 1];          // strictly-lower A_ij (i>j) at off_index(j, i)
    cf u[M], t[M];
    float nu;
    cf ut;
    DS_HD void init(const float* d, const float* o, float diag, const cf* a, const cf* z) {
#pragma unroll
        for (int i = 0; i < M; ++i) { Ad[i] = d[i] + diag; u[i] = a[i]; t[i] = z[i]; }
#pragma unroll
        for (int q = 0; q < M * (M - 1) / 2; ++q) Al[q] = mk(o[2 * q], -o[2 * q + 1]);   // A_ij = conj(R_ji)
        nu = 0.0f;
        ut = mk(0.0f, 0.0f);
        dg = diag;
    }
    DS_HD void column(int j) {
        const float sj = pivot_max(Ad[j], pivot_floor(dg));
#if defined(__HIP_DEVICE_COMPILE__)
        const float r = __builtin_amdgcn_rsqf(sj);     // sj >= 1e-30: a normal number, the bare instruction (what rsqrtf() compiled to while the
                                                       // floor was a literal; against a run-time floor it grew a denormal-range rescale per pivot)
#else
        const float r = 1.0f / sqrtf(sj);
#endif
        const cf uj = cscale(u[j], r), tj = cscale(t[j], r);
        nu = fma_(uj.x, uj.x, fma_(uj.y, uj.y, nu));
        ut = cfmac(ut, tj, uj);                                  // += conj(u_j) t_j
        cf Lc[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i > j) {
                Lc[i] = cscale(Al[off_index(j, i, M)], r);
                u[i] = cfnma(u[i], Lc[i], uj);
                t[i] = cfnma(t[i], Lc[i], tj);
            }
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i > j) {
                Ad[i] = fma_(-Lc[i].x, Lc[i].x, fma_(-Lc[i].y, Lc[i].y, Ad[i]));
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    if (k > j && k < i) {
                        const int q = off_index(k, i, M);
                        Al[q] = cfnmac(Al[q], Lc[i], Lc[k]);             // A_ik -= L_ij conj(L_kj)
                    }
                }
            }
        }
    }
    DS_HD cf finish() const {
        const float inv = rcp_(nu);
        return mk(ut.x * inv, ut.y * inv);
    }
};
// CF: the conjugation of the stored words folded into column 0's instructions (below).  Same results bit for bit; off for the one-pass MVDR +
// post-filter kernel, which it costs 1.6 % at one hop per call (profiles/r05a/mvdr_noconj_ab.txt; + 1.4 % with 10 s per call — the one-hop figure is
// that workload's headline) while the plain MVDR kernel gains in both regimes (+ 0.5 % / + 2 %)
template <int M, bool CF = true> DS_HD cf mvdr_output(const float* d, const float* o, float diag, const cf* a, const cf* z, float floor_) {
    float Ad[M];
    cf Al[M * (M - 1) / 2 + 1];          // strictly-lower A_ij (i>j) at off_index(j, i)
    cf u[M], t[M];
#pragma unroll
    for (int i = 0; i < M; ++i) { Ad[i] = d[i] + diag; u[i] = a[i]; t[i] = z[i]; }
    // A_ij = conj(R_ji): the stored words as they are — column 0 conjugates them inside its scale and its update (cscale_c / cfnmac_ca: the
    // packed instructions' own negations) instead of a v_xor and a v_mov per word into fresh register pairs first
#pragma unroll
    for (int q = 0; q < M * (M - 1) / 2; ++q) Al[q] = CF ? mk(o[2 * q], o[2 * q + 1]) : mk(o[2 * q], -o[2 * q + 1]);
    float nu = 0.0f;
    cf ut = mk(0.0f, 0.0f);
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const float sj = pivot_max(Ad[j], floor_);               // floor_ = pivot_floor(diag)
#if defined(__HIP_DEVICE_COMPILE__)
        const float r = __builtin_amdgcn_rsqf(sj);     // sj >= 1e-30: a normal number, the bare instruction (what rsqrtf() compiled to while the
                                                       // floor was a literal; against a run-time floor it grew a denormal-range rescale per pivot)
#else
        const float r = 1.0f / sqrtf(sj);
#endif
        const cf uj = cscale(u[j], r), tj = cscale(t[j], r);
        nu = fma_(uj.x, uj.x, fma_(uj.y, uj.y, nu));
        ut = cfmac(ut, tj, uj);                                  // += conj(u_j) t_j
        cf Lc[M];
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            Lc[i] = (CF && j == 0) ? cscale_c(Al[off_index(j, i, M)], r) : cscale(Al[off_index(j, i, M)], r);
            u[i] = cfnma(u[i], Lc[i], uj);
            t[i] = cfnma(t[i], Lc[i], tj);
        }
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            Ad[i] = fma_(-Lc[i].x, Lc[i].x, fma_(-Lc[i].y, Lc[i].y, Ad[i]));
#pragma unroll
            for (int k = j + 1; k < i; ++k) {
                const int q = off_index(k, i, M);
                Al[q] = (CF && j == 0) ? cfnmac_ca(Al[q], Lc[i], Lc[k]) : cfnmac(Al[q], Lc[i], Lc[k]);      // A_ik -= L_ij conj(L_kj)
            }
        }
    }
    const float inv = rcp_(nu);
    return mk(ut.x * inv, ut.y * inv);
}
template <int M> DS_HD cf mvdr_output(const float* d, const float* o, float diag, const cf* a, const cf* z) {
    return mvdr_output<M>(d, o, diag, a, z, pivot_floor(diag));
}
#endif

// ---------------------------------------------------------------------------------------------
// Per-bin algorithms.  Return the beamformer output Y[k].
// ---------------------------------------------------------------------------------------------
// Where a lane's Ryy lives when the kernel does NOT hold it in registers across the hops of a call (round 5: the 1024-point kernel with Ryy at
// 6 microphones — a 1024-point workgroup is eight waves, two per SIMD, 256 registers per lane whatever the kernel declares; with Ryy resident
// it carried 60 B of scratch).  Floats [F0, NF) of the lane's state — F0 = the first whole float4 plane behind the MCRA words — stay in
// their HBM planes: every hop reads them, updates them and writes them back (the same recursion on the same values: the same results); the
// few Ryy words in front of F0 share a plane with MCRA's lambda_d and stay in registers.  The TFGSC solve re-reads the columns it needs.
// (At 8 microphones the same arrangement — also plane by plane, also with Rvv parked in its planes around the solves — left 60 to 500 B
// of scratch in every form tried: that kernel and the 8-microphone GSC kernel at 1024 points keep theirs)
struct StreamRef {
    vec4* planes;      // plane F0 / 4 of this lane's bin: &bins[(F0 / 4) * KP + k]; plane q of the streamed part at planes[q * KP]
    float* tail;       // the narrow plane's floats of this bin (NF % 4 of them)
    int KP;
};
template <int M> struct RyyStreamLayout {
    static constexpr int NF = 2 * M * M + 5, RY0 = M * M + 5;
    static constexpr int F0 = (RY0 + 3) & ~3;                      // first streamed float
    static constexpr int NPS = NF / 4 - F0 / 4, RT = NF % 4;       // whole streamed planes, floats in the narrow plane
    static constexpr int HEAD = F0 - RY0;                          // Ryy words that stay in registers (st[RY0 .. F0))
};
template <int M> DS_HD void ryy_stream_load(const StreamRef& sr, const float* st, float* ry) {
    typedef RyyStreamLayout<M> L;
#pragma unroll
    for (int f = 0; f < L::HEAD; ++f) ry[f] = st[L::RY0 + f];
#pragma unroll
    for (int q = 0; q < L::NPS; ++q) {
        const vec4 v = load_state(&sr.planes[q * sr.KP]);
        ry[L::HEAD + 4 * q] = v.x; ry[L::HEAD + 4 * q + 1] = v.y; ry[L::HEAD + 4 * q + 2] = v.z; ry[L::HEAD + 4 * q + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < L::RT; ++j) ry[L::HEAD + 4 * L::NPS + j] = sr.tail[j];
}
template <int M> DS_HD void ryy_stream_store(const StreamRef& sr, float* st, const float* ry) {
    typedef RyyStreamLayout<M> L;
#pragma unroll
    for (int f = 0; f < L::HEAD; ++f) st[L::RY0 + f] = ry[f];
#pragma unroll
    for (int q = 0; q < L::NPS; ++q) {
        vec4 v; v.x = ry[L::HEAD + 4 * q]; v.y = ry[L::HEAD + 4 * q + 1]; v.z = ry[L::HEAD + 4 * q + 2]; v.w = ry[L::HEAD + 4 * q + 3];
        store_state(&sr.planes[q * sr.KP], v);
    }
#pragma unroll
    for (int j = 0; j < L::RT; ++j) sr.tail[j] = ry[L::HEAD + 4 * L::NPS + j];
}
// one word of the streamed Ryy (ry index = float index - RY0) straight from memory
template <int M> DS_HD float ryy_stream_word(const StreamRef& sr, const float* st, int w) {
    typedef RyyStreamLayout<M> L;
    if (w < L::HEAD) return st[L::RY0 + w];
    const int f = w - L::HEAD;
    if (f < 4 * L::NPS) return reinterpret_cast<const float*>(&sr.planes[(f >> 2) * sr.KP])[f & 3];
    return sr.tail[f - 4 * L::NPS];
}

DS_HD bool gate_open(const Params& p, int k) { return (unsigned)k < ~p.gate_kinv; }
template <int M, bool RYY, bool CF = true>
DS_HD cf adaptive_bin(float* st, const cf* Z, const cf* a, const Params& p, const StreamRef* sr = nullptr, int k = 0) {
    typedef StateLayout<M, ALGO_ADAPTIVE, RYY> SL;
    float* d = st + SL::R_DIAG;
    float* o = st + SL::R_OFF;
    if (RYY) {
        if (sr) {                                                                             // streamed: HBM -> update -> HBM, this hop
            float ry[M * M];
            ryy_stream_load<M>(*sr, st, ry);
            herm_rank1<M>(ry, ry + M, Z, p.alpha_y, p.beta_y);
            ryy_stream_store<M>(*sr, st, ry);
        } else {
            herm_rank1<M>(st + SL::RYY_DIAG, st + SL::RYY_OFF, Z, p.alpha_y, p.beta_y);      // :86-88
        }
    }
#ifndef DS_ABLATE_NORANK1      // timing experiment only (scratch/build_variant.sh): the frame program without the covariance accumulate
    if (st[SL::MC_S + 3] < p.gate && gate_open(p, k)) herm_rank1<M>(d, o, Z, p.alpha_v, p.beta_v);   // :90-99
#endif
    cf acc = mk(0.0f, 0.0f);
    if (p.method == METHOD_SRC) {                              // beamformer.py:320-322
        acc = cmulc(Z[0], a[0]);
    } else if (p.method == METHOD_DS) {                        // beamformer.py:323-324
#pragma unroll
        for (int m = 0; m < M; ++m) acc = cfmac(acc, Z[m], a[m]);
        acc = cscale(acc, 1.0f / M);
    } else if (p.method == METHOD_MVDR) {                      // beamformer.py:325-326, :103-104
#ifdef DS_ABLATE_NOSOLVE       // timing experiment only: the frame program without the Hermitian solve
        acc = cmulc(Z[0], a[0]);
#else
#ifdef DS_SOLVE_FP64
        acc = mvdr_output<M>(d, o, p.diag, a, Z);
#else
        acc = mvdr_output<M, CF>(d, o, p.diag, a, Z, p.diag_floor);
#endif
#endif
    } else if (RYY) {                                          // TFGSC, beamformer.py:327-333
        Chol<M> ch;
        ch.factor(d, o, p.diag);
        const float* yd = st + SL::RYY_DIAG;
        const float* yo = st + SL::RYY_OFF;
        cf tr = mk(0.0f, 0.0f);
        cf col0[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            cf b[M], v[M];
#pragma unroll
            for (int i = 0; i < M; ++i) {
                // (element by element, each word pinned to a register of its own: read as (re, im) PAIRS the off-diagonal words — which start
                // at an odd float of the state for even M — became 8-byte accesses across the 16-byte plane groups, and the Ryy planes
                // stayed in private memory: 144 - 416 B of scratch in the 6- and 8-microphone kernels)
                float re, im;
                if (sr) {                                          // streamed Ryy: the words of column j from their planes (written above, this hop)
                    if (i == j) { re = ryy_stream_word<M>(*sr, st, i); im = 0.0f; }
                    else if (i < j) { const int q = off_index(i, j, M); re = ryy_stream_word<M>(*sr, st, M + 2 * q); im = ryy_stream_word<M>(*sr, st, M + 2 * q + 1); }
                    else { const int q = off_index(j, i, M); re = ryy_stream_word<M>(*sr, st, M + 2 * q); im = -ryy_stream_word<M>(*sr, st, M + 2 * q + 1); }
                }
                else if (i == j) { re = yd[i]; im = 0.0f; }
                else if (i < j) { const int q = off_index(i, j, M); re = yo[2 * q]; im = yo[2 * q + 1]; }
                else { const int q = off_index(j, i, M); re = yo[2 * q]; im = -yo[2 * q + 1]; }
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(re), "+v"(im));
#endif
                b[i] = mk(re, im);
            }
            ch.solve(b, v);                                    // column j of Rvv_inv @ Ryy
            tr = cadd(tr, v[j]);
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < M; ++i) col0[i] = v[i];
            }
        }
        col0[0].x -= 1.0f;
        const cf den = mk(tr.x - (float)M, tr.y);
#pragma unroll
        for (int m = 0; m < M; ++m) acc = cfmac(acc, Z[m], cdiv(col0[m], den));
    }
    return acc;
}

template <int M> DS_HD cf fixed_bin(const cf* Z, const cf* w) {   // fixedbeamformer.py:163
    cf acc = mk(0.0f, 0.0f);
#pragma unroll
    for (int m = 0; m < M; ++m) acc = cfmac(acc, Z[m], w[m]);
    return acc;
}

// real symmetric packed helpers (GSC / McMcra) --------------------------------------------------
template <int M> DS_HD float sym_get(const float* s, int i, int j) {
    return i <= j ? s[sym_index(i, j, M)] : s[sym_index(j, i, M)];
}

// McMcra.estimation for one bin (mc_mcra.py:179-224): state pyy/pvv = packed real-symmetric Phi_yy / Phi_vv.
// Returns the speech presence probability p, the gain G and (for state read-back) xi, gamma.
template <int M>
DS_HD void mcmcra_bin(float* pyy, float* pvv, const cf* Z, int k, int spp_frm_cnt, float& p_out, float& G_out,
                      float& xi_out, float& gamma_out) {
    constexpr int NS = M * (M + 1) / 2;
    const float alpha = 0.92f, one_m_alpha = (float)(1.0 - 0.92);
    float yy[NS];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
            const int q = sym_index(i, j, M);
            yy[q] = fma_(Z[i].x, Z[j].x, Z[i].y * Z[j].y);         // Re(conj(y_i) y_j)  :182-184
            pyy[q] = fma_(alpha, pyy[q], one_m_alpha * yy[q]);
        }
    if (spp_frm_cnt < 5) {                                     // :186-187
#pragma unroll
        for (int q = 0; q < NS; ++q) pvv[q] = pyy[q];
    }
    // real Cholesky of Phi_vv + 1e-6 I, then full inverse (needed for the traces)  :191
    float Lm[M][M];
    float inv_d[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        float s = sym_get<M>(pvv, j, j) + 1e-6f;
#pragma unroll
        for (int q = 0; q < j; ++q) s = fma_(-Lm[j][q], Lm[j][q], s);
        s = pivot_max(s, pivot_floor(1e-6f));
        const float r = rsq_(s);
        inv_d[j] = r;
        Lm[j][j] = s * r;
#pragma unroll
        for (int i = j + 1; i < M; ++i) {
            float t = sym_get<M>(pvv, i, j);
#pragma unroll
            for (int q = 0; q < j; ++q) t = fma_(-Lm[i][q], Lm[j][q], t);
            Lm[i][j] = t * r;
        }
    }
    // Linv (lower) by forward substitution on identity columns
    float Li[M][M];
#pragma unroll
    for (int c = 0; c < M; ++c) {
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i < c) { Li[i][c] = 0.0f; continue; }
            float t = (i == c) ? 1.0f : 0.0f;
#pragma unroll
            for (int q = c; q < i; ++q) t = fma_(-Lm[i][q], Li[q][c], t);
            Li[i][c] = t * inv_d[i];
        }
    }
    // inv = Li^T Li (symmetric)
    float iv[NS];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
            float t = 0.0f;
#pragma unroll
            for (int q = j; q < M; ++q) t = fma_(Li[q][i], Li[q][j], t);
            iv[sym_index(i, j, M)] = t;
        }
    // tr = trace(inv Phi_yy), psi = Re(y inv y^H) = sum inv_ij yy_ij, gamma = Re(y^H inv Phi_xx inv y) = v^H Phi_xx v
    // with v = inv y (inv, Phi_* real symmetric): all three are weighted sums over the upper triangle.
    float tr = 0.0f, psi = 0.0f;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
            const int q = sym_index(i, j, M);
            const float e = (i == j) ? iv[q] : 2.0f * iv[q];
            tr = fma_(e, pyy[q], tr);
            psi = fma_(e, yy[q], psi);
        }
    float xi = fminf_(fmaxf_(tr - (float)M, 1e-6f), 1e6f);                               // :193-194
    cf v[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        cf acc = mk(0.0f, 0.0f);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const float e = sym_get<M>(iv, i, j);
            acc.x = fma_(e, Z[j].x, acc.x); acc.y = fma_(e, Z[j].y, acc.y);
        }
        v[i] = acc;
    }
    float gam = 0.0f;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
            const int q = sym_index(i, j, M);
            const float vv = fma_(v[i].x, v[j].x, v[i].y * v[j].y);                       // Re(conj(v_i) v_j)
            gam = fma_((i == j) ? (pyy[q] - pvv[q]) : 2.0f * (pyy[q] - pvv[q]), vv, gam);
        }
    gam = fminf_(fmaxf_(gam, 1e-6f), 1e6f);                                               // :199
    // q_local :91-105
    const float q_max = 0.99f, q_min = 0.01f, psi0 = 100.0f;
    float q;
    if (psi >= psi0 || tr > psi0) q = q_min;
    else if (tr < (float)M) q = q_max;
    else q = fminf_(fmaxf_(div_(psi0 - tr, psi0 - (float)M), q_min), q_max);
    // p :143-151
    const float r1xi = rcp_(1.0f + xi);
    float pp = rcp_(1.0f + div_(q, 1.0f - q) * (1.0f + xi) * exp_(-1.0f * (gam * r1xi)));
    pp = fminf_(fmaxf_(pp, 0.01f), 0.99f);
    // noise PSD update :210-224
    const float at = 0.95f + (float)(1.0 - 0.95) * pp;
#pragma unroll
    for (int q2 = 0; q2 < NS; ++q2) pvv[q2] = fma_(at, pvv[q2], (1.0f - at) * yy[q2]);
    // gain :153-157
    const float Gmin = 0.0631f;
    const float gh1 = xi * r1xi;
    float G = exp2_(fma_(pp, log2_(gh1), (1.0f - pp) * log2f(Gmin)));                      // G_H1^p Gmin^(1-p)
    G = fmaxf_(fminf_(G, 1.0f), Gmin);
    if (k < 2) G = 0.0f;
    p_out = pp; G_out = G; xi_out = xi; gamma_out = gam;
}

// GSC.process for one bin: McMcra SPP/gain (GSC.py:225) + FD-GSC with SPP-stepped LMS canceller (GSC.py:245-286).
template <int M>
DS_HD cf gsc_bin(float* st, const cf* Z, const cf* a, const Params& p, int k, int spp_frm_cnt, long long pw_row0 = -1) {
    typedef StateLayout<M, ALGO_GSC, false> SL;
    float* ga = st + SL::GA;
    float pp, G, xi, gam;
    mcmcra_bin<M>(st + SL::PYY, st + SL::PVV, Z, k, spp_frm_cnt, pp, G, xi, gam);
    if (p.method == 0) return Z[0];                                                        // GSC.py:242-243
    // FD-GSC: W = a/(a^H a), U_i = conj(a_0) z_0 - conj(a_{i+1}) z_{i+1}                   GSC.py:219-222,261-266
    float aa = 0.0f;
    cf yf = mk(0.0f, 0.0f);
#pragma unroll
    for (int m = 0; m < M; ++m) { aa += cabs2(a[m]); yf = cfmac(yf, Z[m], a[m]); }
    yf = cscale(yf, rcp_(aa));
    const cf u0 = cmulc(Z[0], a[0]);
    cf U[M - 1];
    cf Yk = yf;
#pragma unroll
    for (int i = 0; i < M - 1; ++i) {
        U[i] = csub(u0, cmulc(Z[i + 1], a[i + 1]));
        Yk = cfnmac(Yk, U[i], mk(ga[2 * i], ga[2 * i + 1]));                           // conj(G_i) U_i
    }
    if (p.ref_pow != nullptr && pw_row0 >= 0) {                                             // GSC.py:281-283: the powers omlsa_multi.estimation is given
        float* pw = p.ref_pow + (pw_row0 + k) * M;                                          // (a uniform branch; the address only inside it)
        pw[0] = cabs2(Yk);
#pragma unroll
        for (int i = 0; i < M - 1; ++i) pw[1 + i] = cabs2(U[i]);
    }
    const float step = p.mu * (1.0f - pp);                                                  // GSC.py:270-274
#pragma unroll
    for (int i = 0; i < M - 1; ++i) {
        const cf g = cmulc(U[i], Yk);                                                       // U_i conj(Y)
        ga[2 * i] = fma_(step, g.x, ga[2 * i]);
        ga[2 * i + 1] = fma_(step, g.y, ga[2 * i + 1]);
    }
    return cscale(Yk, G);                                                                   // GSC.py:286
}

// The SubbandGSC canceller for one bin: SubbandLmsMc.update with 2 taps on M channels (SubbandLmsMc.py:62-63,150-190 through
// SubbandGSC.py:230-236), the arithmetic of op_sublms_t<2, M> (ds_ops.hpp) word for word.  st = [W (2 M complex) | X (2 M complex) | P].
template <int M> DS_HD cf aic_bin(float* st, const cf* Xin, cf d, float pk, const Params& p) {
    constexpr int NC = 2 * M, NC2 = 2 * NC;
    float* W = st;
    float* X = st + NC2;
#pragma unroll
    for (int c = 0; c < M; ++c) {
        X[2 * (M + c)] = X[2 * c]; X[2 * (M + c) + 1] = X[2 * c + 1];            // tap 1 <- tap 0
        X[2 * c] = Xin[c].x; X[2 * c + 1] = Xin[c].y;                              // tap 0 <- this frame
    }
    cf out = mk(0.0f, 0.0f);
    float pw = 0.0f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const cf x = mk(X[2 * i], X[2 * i + 1]);
        out = cfmac(out, x, mk(W[2 * i], W[2 * i + 1]));                           // conj(W) X
        pw += cabs2(x);
    }
    if (p.aic_pc) pk = 1.0f - pk;
    const cf err = mk(fma_(-out.x, pk, d.x), fma_(-out.y, pk, d.y));             // d - out * p
    float scale = 1.0f;
    if (p.aic_norm) {
        float P = st[2 * NC2];
        P = fma_(p.aic_alpha, P, (1.0f - p.aic_alpha) * (pw / (float)M));
        st[2 * NC2] = P;
        scale = 1.0f / (P + p.aic_reg);
    }
    const float g = 2.0f * p.aic_mu * pk * scale;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const cf gr = cmulc(mk(X[2 * i], X[2 * i + 1]), err);                     // X conj(err)
        W[2 * i] = fma_(g, gr.x, W[2 * i]);
        W[2 * i + 1] = fma_(g, gr.y, W[2 * i + 1]);
    }
    return err;
}
// plane f of the canceller's state at bin k of utterance b.  On the device one buffer descriptor per utterance, the lane's bin as the
// 32-bit offset and the plane in the scalar offset (the addressing of the operator kernels, ds_ops.hpp)
// (the operator kernels' layout, ds_ops.hpp st_index: float f of bin k at [f / 4][k][f % 4], NF rounded up to whole float4 planes)
DS_HD constexpr int aic_floats_per_bin(int NF) { return (NF + 3) & ~3; }
#if defined(__HIP_DEVICE_COMPILE__)
struct AicState {
    __amdgpu_buffer_rsrc_t rs;
    int KP;
    __device__ AicState(const Params& p, int b, int KP_) : KP(KP_) {
        rs = __builtin_amdgcn_make_buffer_rsrc(p.aic_st + (long long)b * aic_floats_per_bin(p.aic_NF) * KP_, 0, aic_floats_per_bin(p.aic_NF) * KP_ * 4, 0x00020000);
    }
    __device__ float ld(int f, int k) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((k * 4 + (f & 3)) * 4), (unsigned)((f >> 2) * KP * 16), 0)); }
    __device__ void st(int f, int k, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (unsigned)((k * 4 + (f & 3)) * 4), (unsigned)((f >> 2) * KP * 16), 0); }
    // the lanes' own planes, once in and once out per launch: streamed (cache policy 2 = nt), like the operators' state (ds_ops.hpp)
#if defined(DS_PLAIN_STATE) || defined(DS_AIC_PLAIN_STATE)
    __device__ float ld_once(int f, int k) const { return ld(f, k); }
    __device__ void st_once(int f, int k, float v) const { st(f, k, v); }
#else
    __device__ float ld_once(int f, int k) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((k * 4 + (f & 3)) * 4), (unsigned)((f >> 2) * KP * 16), 2)); }
    __device__ void st_once(int f, int k, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, (unsigned)((k * 4 + (f & 3)) * 4), (unsigned)((f >> 2) * KP * 16), 2); }
#endif
};
#else
struct AicState {
    float* base;
    int KP;
    AicState(const Params& p, int b, int KP_) : base(p.aic_st ? p.aic_st + (long long)b * aic_floats_per_bin(p.aic_NF) * KP_ : nullptr), KP(KP_) {}
    float ld(int f, int k) const { return base[((long long)(f >> 2) * KP + k) * 4 + (f & 3)]; }
    void st(int f, int k, float v) const { base[((long long)(f >> 2) * KP + k) * 4 + (f & 3)] = v; }
    float ld_once(int f, int k) const { return ld(f, k); }
    void st_once(int f, int k, float v) const { st(f, k, v); }
};
#endif

// ---------------------------------------------------------------------------------------------
// The block program
// ---------------------------------------------------------------------------------------------
template <int NFFT, int M, int ALGO, bool RYY> struct Engine {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2;
    static constexpr int NT = NC;                            // one thread per bin 0..NC-1
    // 512-point frames: every stage is radix-4 with exactly 64 butterflies per channel, so a channel's transform never leaves its
    // wavefront (the 128- and 512-point plans end in a radix-2 stage with a different butterfly-to-lane map and keep their barriers)
    static constexpr bool WAVE_FFT = NC == 256;
    static constexpr int NYQ_TID = 0;                        // lane (wave 0, idle during the inverse FFT stages) that runs the Nyquist bin k = NC
    static constexpr int INV_T0 = NT / 2;                    // inverse-FFT butterflies run on threads [NT/2, NT)
    static constexpr int KP = plane_len(K);                  // padded plane length
    typedef StateLayout<M, ALGO, RYY> SL;
    static constexpr int NP = SL::NP;
    // the 1024-point kernel with Ryy at 6 microphones streams Ryy through its HBM planes hop by hop instead of holding it (StreamRef)
    static constexpr bool STREAM_RYY = RYY && ALGO == ALGO_ADAPTIVE && NFFT >= 1024 && M == 6;
    static constexpr int NPLD = STREAM_RYY ? RyyStreamLayout<M>::F0 / 4 : SL::NPF;      // float4 planes a lane keeps in registers for the call
    static constexpr int RTLD = STREAM_RYY ? 0 : SL::RT;                                // ... and floats of the narrow plane
    static constexpr int NV4 = HOP * M / 4;                  // float4 per hop of input
    static constexpr int NPRE = (NV4 + NT - 1) / NT;
    typedef Shared<NFFT, M, (SL::NP > 0 ? SL::NP * 4 : 4)> Sh;
    typedef Regs<M, ALGO, RYY, NPRE> Rg;
    static constexpr bool FWD_FINAL_IS_FB = (NC != 512);     // 128: 4 stages, 256: 4 stages, 512: 5 stages
    // 8 microphones at 1024 points (one workgroup of eight waves per CU, 256 registers per lane and all of them in use): the next hop's input
    // is fetched while the inverse transform runs instead of at the top of the hop (its 2 x 4 values and two addresses are then not live
    // across the per-bin program), and the lane that runs the Nyquist bin's pass parks part of its own bin's state in the transform
    // buffer that is idle during the per-bin phase — the two places where this shape left values in scratch (124 B)
    // (GSC only: measured on the other 8-microphone 1024-point kernels, which had no scratch to lose, the late staging costs 4 - 8 % with 40 hops
    // per call; the GSC kernel gains 10 - 12 % at one hop per call and at 40: profiles/r05a/m8_1024_late_ab.txt)
    static constexpr bool LATE_PREFETCH = NFFT >= 1024 && M >= 8 && ALGO == ALGO_GSC;
    static constexpr int NYQ_PARK = (NFFT >= 1024 && M >= 8 && ALGO == ALGO_GSC) ? M * (M + 1) / 2 : 0;     // floats st[PVV ..) parked
    static constexpr bool INV_FINAL_IS_FA = (NC != 512);

    // this lane's first input address (hop 0); the stream is then walked by pointer increments
    static DS_HD void prefetch_init(const Params& p, long long xb, int tid, Rg& r) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int v = tid + i * NT;
            long long off = xb;
            if (v < NV4) {
                if (p.x_sample_stride == 1) {                // [M][L]
                    const int m = v / (HOP / 4), q = v - m * (HOP / 4);
                    off = xb + (long long)m * p.x_chan_stride + 4 * q;
                } else {                                     // [L][M] interleaved
                    off = xb + 4 * v;
                }
            }
            r.xp[i] = reinterpret_cast<const vec4*>(p.x + off);
        }
    }
    // issue the global loads of the next hop (all channels) into registers
    static DS_HD void prefetch(const Params& p, long long xb, int t, int tid, Rg& r) {
        const int step = p.x_sample_stride == 1 ? HOP / 4 : HOP * M / 4;   // vec4 per hop along this lane's stream
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            if (tid + i * NT < NV4) {
#if defined(__HIP_DEVICE_COMPILE__)
                // the input stream is read once: non-temporal, so that it does not evict state lines on their way in / out
                // (measured +4..5 % at B = 1024, one hop per call; neutral in the chunked regime)
                typedef float f4_t __attribute__((ext_vector_type(4)));
                const f4_t q = __builtin_nontemporal_load(reinterpret_cast<const f4_t*>(r.xp[i]));
                r.pre[i].x = q.x; r.pre[i].y = q.y; r.pre[i].z = q.z; r.pre[i].w = q.w;
#else
                r.pre[i] = *r.xp[i];
#endif
                r.xp[i] += step;
            }
        }
    }
    // registers -> LDS new half
    static DS_HD void commit(const Params& p, Sh& sh, int new_half, int tid, const Rg& r) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int v = tid + i * NT;
            if (v < NV4) {
                const float e[4] = {r.pre[i].x, r.pre[i].y, r.pre[i].z, r.pre[i].w};
                if (p.x_sample_stride == 1) {
                    const int m = v / (HOP / 4), q = v - m * (HOP / 4);
                    *reinterpret_cast<vec4*>(&sh.xbuf[m][new_half * HOP + 4 * q]) = r.pre[i];   // one 16-byte LDS store
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int lin = 4 * v + c, n = lin / M, m = lin - n * M;
                        sh.xbuf[m][new_half * HOP + n] = e[c];
                    }
                }
            }
        }
    }

    // one frequency bin: MCRA / covariance recursion / solve -> Y[k]
    static DS_HD cf bin_program(float* st, const cf* Z, const cf* steer, int k, const Sh& sh, const Params& p,
                                int frm_cnt, bool reset, int spp_cnt, cf ad = cf{0.0f, 0.0f}, float apk = 0.0f, long long pw_row0 = -1,
                                const StreamRef* sr = nullptr) {
        if constexpr (ALGO == ALGO_AIC) return aic_bin<M>(st, Z, ad, apk, p);
        cf a[M];
#pragma unroll
        for (int m = 0; m < M; ++m) a[m] = steer[k * M + m];
        cf Yk;
        if constexpr (ALGO == ALGO_FIXED) {
            Yk = fixed_bin<M>(Z, a);
        } else if constexpr (ALGO == ALGO_ADAPTIVE) {
            mcra_bin(st + SL::MC_S, k, K, sh.pw[k > 0 ? k - 1 : 0], sh.pw[k], sh.pw[k + 1], frm_cnt, reset, p.mcra_L);
            Yk = adaptive_bin<M, RYY>(st, Z, a, p, sr, k);
        } else if constexpr (ALGO == ALGO_ADAPTIVE_PF) {
            mcra_bin(st + SL::MC_S, k, K, sh.pw[k > 0 ? k - 1 : 0], sh.pw[k], sh.pw[k + 1], frm_cnt, reset, p.mcra_L);
#if defined(DS_PF_LONG)          // (the long-call build of this kernel: ds_kernels_adaptive_pf_long.hip)
            Yk = adaptive_bin<M, false, true>(st, Z, a, p, nullptr, k);
#else
            Yk = adaptive_bin<M, false, false>(st, Z, a, p, nullptr, k);                                   // adaptivebeamformer.py:69-120
#endif
            float pp, G, xi, gam;
            mcmcra_bin<M>(st + SL::PF_PYY, st + SL::PF_PVV, Z, k, spp_cnt, pp, G, xi, gam);    // spp.estimation(Z)  GSC.py:225
            Yk = cscale(Yk, G);                                                                // Y * spp.G         GSC.py:286
        } else {
            Yk = gsc_bin<M>(st, Z, a, p, k, spp_cnt, pw_row0);
        }
        return Yk;
    }

    template <class Exec> static DS_HD void run(Exec& ex, const Params& p, int blk, Sh& sh) {
        const int b = p.batch0 + blk;
        const long long xb = (long long)blk * p.x_batch_stride;
        const long long yb = (long long)blk * p.y_batch_stride;
        float* const ubase = reinterpret_cast<float*>(p.bins) + (long long)b * SL::ust(KP);      // this utterance's state
        vec4* bins = reinterpret_cast<vec4*>(ubase);                                            // its NPF full planes ...
        float* const btail = ubase + (long long)SL::NPF * KP * 4;                               // ... and the narrow one, [KP][RT]
        float* tin = p.tail_in + (long long)b * M * HOP;
        float* tout = p.tail_out + (long long)b * HOP;
        int* cnt = p.counters + (long long)b * 4;
        const cf* steer = p.steer + (long long)b * p.steer_batch_stride;
        // (the fixed beamformer and the chain tail have no frame counters: nothing to read, nothing to write back — as loaded-and-restored
        // values they were three registers held across the whole kernel, spilled to scratch in the 6-microphone tail)
        constexpr bool HAS_CNT = algo_has_mcra(ALGO) || algo_has_spp(ALGO);
        int frm_cnt = HAS_CNT ? cnt[0] : 0, ell = HAS_CNT ? cnt[1] : 1, spp_cnt = HAS_CNT ? cnt[2] : 0;
        int old_half = 0;
        // Params::ref_pow: row of frame t, bin 0 of this utterance in [B][T][K] (GSC only; wave-uniform)
        auto ref_pow_row = [&](int t) -> long long { return ALGO == ALGO_GSC ? ((long long)b * p.T + t) * K : -1; };
#if defined(DS_LATE_STATE_STORE)
        constexpr bool EARLY_STORE = false;                        // A/B switch: everything in the epilogue, as before round 3
#else
        constexpr bool EARLY_STORE = ALGO != ALGO_AIC;
#endif
        auto store_nyquist_state = [&](int tid) {                  // the Nyquist bin's planes (kept in LDS during the call)
            DS_PIN(tid);
            if (tid < SL::NPF) {
                vec4 v; v.x = sh.nyq[4 * tid]; v.y = sh.nyq[4 * tid + 1]; v.z = sh.nyq[4 * tid + 2]; v.w = sh.nyq[4 * tid + 3];
                bins[tid * KP + NC] = v;
            } else if (SL::RT > 0 && tid == SL::NPF) {
#pragma unroll
                for (int j = 0; j < SL::RT; ++j) btail[NC * SL::RT + j] = sh.nyq[4 * SL::NPF + j];
            }
        };
        auto store_lane_state = [&](int tid, Rg& r) {              // bin `tid`'s planes back to HBM
            DS_PIN(tid);                                           // addresses formed here, not hoisted out of the hop loop (registers)
#pragma unroll
            for (int q = 0; q < NPLD; ++q) {
                vec4 v; v.x = r.st[4 * q]; v.y = r.st[4 * q + 1]; v.z = r.st[4 * q + 2]; v.w = r.st[4 * q + 3];
                store_state(&bins[q * KP + tid], v);
            }
#pragma unroll
            for (int j = 0; j < RTLD; ++j) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(DS_NT_TAIL)
                __builtin_nontemporal_store(r.st[4 * SL::NPF + j], &btail[tid * SL::RT + j]);
#else
                btail[tid * SL::RT + j] = r.st[4 * SL::NPF + j];
#endif
            }
        };
        // Wave priorities.  The per-bin phase is wide and arithmetic-heavy, everything else in a hop is a chain of short LDS round trips.  When
        // the whole grid is resident at once (4 workgroups per CU for 4 microphones and 512-point frames) the chains run at raised priority from the start of a hop and the per-bin
        // phase yields to them (+6..9 % at B = 1024, one hop per call; +8 % chunked); with more workgroups than that waiting for a slot only
        // the serial tail of a hop (inverse stages, Nyquist bin, overlap-add) is raised (the full scheme costs 1.5 % there).
        constexpr int by_lds = 160 * 1024 / (int)sizeof(Sh), by_threads = 2048 / NT;
        constexpr int resident = 256 * (by_lds < by_threads ? by_lds : by_threads);      // workgroups the 256 CUs hold at once
        const bool one_round = DS_GRID_BLOCKS() <= resident;

        // LATE_PREFETCH: hop `t` of the input straight into half `half` of the sample buffer.  [M][L] streams go global -> LDS with no register in
        // between (a wavefront's 64 x 16 bytes are 256 consecutive samples of one channel); the interleaved layout loads and scatters on the spot
        auto stage_hop_late = [&](int t, int half, int tid, Rg& r) {
            if (p.x_sample_stride == 1) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int v = tid + i * NT;
                    if (v < NV4) {
                        const int m = v / (HOP / 4), q = v - m * (HOP / 4), lane = tid & 63;
                        ex.lds_load16(&sh.xbuf[m][half * HOP + 4 * (q - lane)], lane, p.x + xb + (long long)m * p.x_chan_stride + 4 * q + (long long)t * HOP);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int v = tid + i * NT;
                    if (v < NV4) {
                        const vec4 q4 = *reinterpret_cast<const vec4*>(p.x + xb + 4 * v + (long long)t * HOP * M);
                        const float e[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int lin = 4 * v + c, n = lin / M, m = lin - n * M;
                            sh.xbuf[m][half * HOP + n] = e[c];
                        }
                    }
                }
            }
        };
        // ---- prologue: tables, tails, per-bin state ---------------------------------------------
        ex.phase([&](int tid, Rg& r) {
            {   // tables, STFT tail (-> old half 0) and OLA tail: 16-byte copies
                vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
                for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
                const vec4* tin4 = reinterpret_cast<const vec4*>(tin);
                for (int i = tid; i < M * HOP / 4; i += NT) {
                    const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                    *reinterpret_cast<vec4*>(&sh.xbuf[m][4 * q]) = tin4[i];
                }
                const vec4* tout4 = reinterpret_cast<const vec4*>(tout);
                for (int i = tid; i < HOP / 4; i += NT) *reinterpret_cast<vec4*>(&sh.tail[4 * i]) = tout4[i];
            }
            bool spectra_in = false;
            if constexpr (ALGO == ALGO_AIC) spectra_in = p.aic_e != nullptr;
            if (!spectra_in) {
                if constexpr (LATE_PREFETCH) {
                    if (p.T > 0) stage_hop_late(0, 1, tid, r);
                } else {
                    prefetch_init(p, xb, tid, r);
                    prefetch(p, xb, 0, tid, r);
                }
            } else if (tid < NC / 2) {                                  // this lane's samples of the blocking-matrix overlap tails
                const float* bt = p.aic_bmtail + (long long)b * M * HOP;
#pragma unroll
                for (int c = 0; c < M; ++c) { r.bmt[2 * c] = bt[c * HOP + 2 * tid]; r.bmt[2 * c + 1] = bt[c * HOP + 2 * tid + 1]; }
            }
            // state planes are issued LAST and consumed first in the per-bin phase: they stay in flight while the
            // forward FFT of the first hop runs (loads retire in order, so nothing above waits for them)
            if constexpr (ALGO == ALGO_AIC) {                           // the canceller's planes, where the subband-LMS operator keeps them
                const AicState as(p, b, KP);
#pragma unroll
                for (int f = 0; f < SL::NF; ++f) r.st[f] = as.ld_once(f, tid);
                r.nyq.x = as.ld(tid < SL::NF ? tid : 0, NC);
            } else {
#pragma unroll
                for (int q = 0; q < NPLD; ++q) {
                    const vec4 v = load_state(&bins[q * KP + tid]);
                    r.st[4 * q] = v.x; r.st[4 * q + 1] = v.y; r.st[4 * q + 2] = v.z; r.st[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < RTLD; ++j) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(DS_NT_TAIL)
                    r.st[4 * SL::NPF + j] = __builtin_nontemporal_load(&btail[tid * SL::RT + j]);
#else
                    r.st[4 * SL::NPF + j] = btail[tid * SL::RT + j];
#endif
                }
                if constexpr (NP > 0) {                                     // Nyquist planes: parked in a register until the split phase
                    r.nyq = bins[(tid < SL::NPF ? tid : 0) * KP + NC];
                    if (SL::RT > 0 && tid == SL::NPF) {
                        float w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                        for (int j = 0; j < SL::RT; ++j) w[j] = btail[NC * SL::RT + j];
                        r.nyq.x = w[0]; r.nyq.y = w[1]; r.nyq.z = w[2]; r.nyq.w = w[3];
                    }
                }
            }
            if constexpr (LATE_PREFETCH) ex.lds_load_wait();            // hop 0 has landed (and, with it, the state planes: this shape gives up that overlap)
        });
        // ALGO_AIC: this frame's desired-signal sample (the spectrum one frame back; frame 0 takes the carried one) and update
        // probability for the lane's bin, and for the Nyquist bin on the lane that runs it; issued at the start of a frame, consumed
        // behind the forward transforms
        auto aic_fetch = [&](int t, int tid, Rg& r) {
            auto dsamp = [&](int k) {
                const float* q = t == 0 ? p.aic_dprev + 2 * ((long long)b * K + k) : p.aic_d + 2 * (((long long)b * p.T + (t - 1)) * K + k);
                return mk(q[0], q[1]);
            };
            auto prob = [&](int k) { return p.aic_p ? p.aic_p[((long long)b * p.T + t) * K + k] : 1.0f; };
            r.ad = dsamp(tid); r.apk = prob(tid);
            if (tid == NYQ_TID) { r.adn = dsamp(NC); r.apkn = prob(NC); }
        };

#ifdef DS_ABLATE_NOFRAMES     // timing experiment only: state movement without the frame program
        const int T_run = 0;
#else
        const int T_run = p.T;
#endif
        for (int t = 0; t < T_run; ++t) {
            const int new_half = old_half ^ 1;
            // ---- hop t into LDS, start fetching hop t+1 ------------------------------------------
            // WAVE_FFT: a channel's transform lives inside one wavefront from the staging of its samples to the last stage, so those
            // hand-offs need no workgroup barrier (the interleaved input layout scatters the staging across channels and keeps its barrier).
            auto ph = [&](bool wave_local, auto f) { if (wave_local) ex.phase_wave(f); else ex.phase(f); };
            auto phs = [&](bool wave_local, auto f) { if (wave_local) ex.stage_wave(f); else ex.stage(f); };     // transform stages
            cf* fa = &sh.fa[0][0];
            cf* fb = &sh.fb[0][0];
            bool spectra_in = false;
            if constexpr (ALGO == ALGO_AIC) spectra_in = p.aic_e != nullptr;
            if (spectra_in) {
                // ---- hop t of the M blocking-matrix outputs from the error spectra: IstftEngine's program (merge, inverse stages, window,
                // overlap-add) with the new samples going to the re-analysis through LDS instead of HBM ------------------------------------
                ex.phase([&](int tid, Rg& r) {
                    if (one_round) DS_SETPRIO(2);
                    if constexpr (ALGO == ALGO_AIC) aic_fetch(t, tid, r);
                    // (these 2 M loads open every frame.  Round 6 issued them one frame ahead, behind the output's inverse transform — 24 more
                    // words in flight, no more registers at the kernel's peak: cfg5 26.6 - 26.9 M against 27.1 - 27.3 M frames/s with 10 s per
                    // call, no change per block.  The other two workgroups of the CU already cover the wait.  profiles/r06a/tail_prefetch_ab.txt)
                    const int k = tid;
                    const cf w = cconj(sh.tb.tw[k]);
#pragma unroll
                    for (int c = 0; c < M; ++c) {
                        const cf* Yt = reinterpret_cast<const cf*>(p.aic_e) + (((long long)b * M + c) * p.T + t) * K;
                        cf A = Yt[k], B = Yt[NC - k];
                        if (k == 0) { A.y = 0.0f; B.y = 0.0f; }          // irfft ignores Im Y[0], Im Y[N/2]
                        const cf E = cscale(cadd_c(A, B), 0.5f);
                        const cf O = cmul(cscale(csub_c(A, B), 0.5f), w);
                        fa[c * Sh::NCP + k] = cadd_jd<+1>(E, O);
                    }
                });
                ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 0, 1>(tid, NT, sh, fa, fb, 1, 0, M); });
                ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 1, 2>(tid, NT, sh, fb, fa, 4, 0, M); });
                ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 2, 0>(tid, NT, sh, fa, fb, 16, 0, M); });
                if (NC == 128) {
                    ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid, NT, sh, fb, fa, 64, 0, M); });
                } else {
                    ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 0, 0>(tid, NT, sh, fb, fa, 64, 0, M); });
                    if (NC == 512)
                        ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid, NT, sh, fa, fb, 256, 0, M); });
                }
                const cf* Zb = (NC != 512) ? fa : fb;
                ex.phase([&](int tid, Rg& r) {
                    if (tid < NC / 2) {
                        const int i = tid;
                        const float sc = 1.0f / (float)NC;
#pragma unroll
                        for (int c = 0; c < M; ++c) {
                            const cf z1 = Zb[c * Sh::NCP + i], z2 = Zb[c * Sh::NCP + i + NC / 2];
                            const float y0 = sh.tb.win[2 * i] * (z1.x * sc), y1 = sh.tb.win[2 * i + 1] * (z1.y * sc);
                            const float o0 = (y0 + r.bmt[2 * c]) * p.out_scale, o1 = (y1 + r.bmt[2 * c + 1]) * p.out_scale;
                            r.bmt[2 * c] = sh.tb.win[HOP + 2 * i] * (z2.x * sc);
                            r.bmt[2 * c + 1] = sh.tb.win[HOP + 2 * i + 1] * (z2.y * sc);
                            sh.xbuf[c][new_half * HOP + 2 * i] = o0;
                            sh.xbuf[c][new_half * HOP + 2 * i + 1] = o1;
                            if (p.aic_bm) {
                                float* dst = p.aic_bm + ((long long)b * M + c) * p.T * HOP + (long long)t * HOP + 2 * i;
                                dst[0] = o0; dst[1] = o1;
                            }
                        }
                    }
                });
            } else if (!LATE_PREFETCH) {                                // (LATE_PREFETCH: hop t went into LDS in the prologue / in hop t - 1's overlap-add phase)
                ph(WAVE_FFT && p.x_sample_stride == 1, [&](int tid, Rg& r) {
                    if (one_round) DS_SETPRIO(2);
                    commit(p, sh, new_half, tid, r);
                    if (!LATE_PREFETCH && t + 1 < p.T) prefetch(p, xb, t + 1, tid, r);
                    if constexpr (ALGO == ALGO_AIC) aic_fetch(t, tid, r);
                });
            }
            // ---- forward FFT: M packed real transforms -------------------------------------------
            phs(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, true, 0, 1>(tid, NT, sh, nullptr, fa, 1, old_half, M); });
            phs(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 1, 2>(tid, NT, sh, fa, fb, 4, 0, M); });
            phs(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 2, 0>(tid, NT, sh, fb, fa, 16, 0, M); });
            if (NC == 128) {
                ex.stage([&](int tid, Rg&) { fft_stage<NFFT, M, 2, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, M); });
            } else {
                ex.stage([&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, M); });
                if (NC == 512)
                    ex.stage([&](int tid, Rg&) { fft_stage<NFFT, M, 2, -1, false, 0, 0>(tid, NT, sh, fb, fa, 256, 0, M); });
            }
            const cf* F = FWD_FINAL_IS_FB ? fb : fa;
            // ---- split packed spectrum -> Z[k][m]; publish |Z_0|^2 for the MCRA stencil ----------
            ex.phase([&](int tid, Rg& r) {
                if constexpr (ALGO == ALGO_AIC) {
                    if (t == 0 && tid < SL::NF) sh.nyq[tid] = r.nyq.x;
                } else if (t == 0 && tid < NP) {                        // Nyquist planes -> LDS (loaded in the prologue)
                    sh.nyq[4 * tid] = r.nyq.x; sh.nyq[4 * tid + 1] = r.nyq.y; sh.nyq[4 * tid + 2] = r.nyq.z; sh.nyq[4 * tid + 3] = r.nyq.w;
                }
                const int k = tid, k2 = (NC - k) & (NC - 1);
                const cf w = sh.tb.tw[k];
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const cf A = F[m * Sh::NCP + k], B = F[m * Sh::NCP + k2];
                    const cf E = cscale(cadd_c(A, B), 0.5f);
                    const cf D = csub_c(A, B);
                    const cf O = cdiv_2j(D);          // D / (2j)
                    r.Z[m] = cfma(E, w, O);
                }
                if (k == 0) {
#pragma unroll
                    for (int m = 0; m < M; ++m) r.Z[m].y = 0.0f;
                    const cf F0 = F[0];
                    const float zn = F0.x - F0.y;                       // channel-0 Nyquist bin
                    sh.pw[NC] = zn * zn;
                }
                sh.pw[k] = cabs2(r.Z[0]);
            });
            // ---- per-bin recursion -> Y[k] --------------------------------------------------------
            const bool reset = (frm_cnt != 0) && (ell % p.mcra_L == 0);
            ex.phase([&](int tid, Rg& r) {
                DS_SETPRIO(0);                                          // the wide, arithmetic-heavy phase yields to other workgroups' latency-bound ones
                StreamRef srf; const StreamRef* sr = nullptr;
                if constexpr (STREAM_RYY) {
                    srf.planes = &bins[NPLD * KP + tid]; srf.tail = &btail[tid * SL::RT]; srf.KP = KP;
                    sr = &srf;
                }
                cf Yk = bin_program(r.st, r.Z, steer, tid, sh, p, frm_cnt, reset, spp_cnt, r.ad, r.apk, ref_pow_row(t), sr);
                if (tid == 0) Yk.y = 0.0f;                              // irfft ignores Im Y[0] and Im Y[N/2]
                sh.Y[tid] = Yk;
                if constexpr (EARLY_STORE) {
                    // the call's last hop: this lane's state and the analysis overlap are final here — on their way to HBM while the inverse
                    // transform and the overlap-add run, instead of behind them (at one hop per call every launch ends on this)
                    if (t == T_run - 1) {
                        store_lane_state(tid, r);
                        vec4* tin4 = reinterpret_cast<vec4*>(tin);
                        for (int i = tid; i < M * HOP / 4; i += NT) {
                            const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                            store_state(&tin4[i], *reinterpret_cast<const vec4*>(&sh.xbuf[m][new_half * HOP + 4 * q]));
                        }
                    }
                }
                if (tid == NYQ_TID) {
                    if constexpr (WAVE_FFT) {                           // the Nyquist bin's inputs, before the inverse transform reuses the buffer
#pragma unroll
                        for (int m = 0; m < M; ++m) { const cf F0 = F[m * Sh::NCP]; sh.zn[m] = F0.x - F0.y; }
                    } else {                                            // second pass: the Nyquist bin, state in LDS (see below)
                        cf Zn[M];
#pragma unroll
                        for (int m = 0; m < M; ++m) { const cf F0 = F[m * Sh::NCP]; Zn[m] = mk(F0.x - F0.y, 0.0f); }
                        float* const park = reinterpret_cast<float*>(FWD_FINAL_IS_FB ? fa : fb) + 2 * Sh::NCP;     // (row 1 of the idle buffer)
                        if constexpr (NYQ_PARK > 0) {
#pragma unroll
                            for (int j = 0; j < NYQ_PARK; ++j) park[j] = r.st[SL::PVV + j];
                            DS_COMPILER_FENCE();
                        }
#ifdef DS_ABLATE_NYQUIST   // timing experiment only (profiles/r04a/nyquist_ablation.txt): the hop without the Nyquist bin's per-bin program
                        const cf Yn = Zn[0];
#else
                        const cf Yn = bin_program(sh.nyq, Zn, steer, NC, sh, p, frm_cnt, reset, spp_cnt, r.adn, r.apkn, ref_pow_row(t));
#endif
                        sh.Y[NC] = mk(Yn.x, 0.0f);
                        if constexpr (NYQ_PARK > 0) {
                            DS_COMPILER_FENCE();
#pragma unroll
                            for (int j = 0; j < NYQ_PARK; ++j) r.st[SL::PVV + j] = park[j];
                        }
                    }
                }
            });
            const int frm_nyq = frm_cnt, spp_nyq = spp_cnt;
            if (algo_has_mcra(ALGO)) {
                if (reset) ell = 0;
                frm_cnt += 1; ell += 1;
            }
            if (algo_has_spp(ALGO)) spp_cnt += 1;
            // ---- inverse packed real FFT -----------------------------------------------------------
            // The Nyquist bin k = NC is the 257th bin of 256 lanes.  In 512-point frames (wave-local inverse stages) its per-bin program
            // (state in LDS) runs on lane NYQ_TID while another wave runs the inverse FFT stages, instead of as a second pass of the per-bin
            // phase that the whole workgroup waits for (which is what the other frame sizes, with a barrier behind every stage, keep doing).  The transform is linear,
            // so it is taken with Y[NC] = 0 and the bin's contribution — the constant (Y[NC] / 2) (1 - j) on every packed point, i.e.
            // +- Y[NC] / 2 on even / odd samples — is added in the overlap-add phase.
            // The merge of the packed real transform (Y[k], Y[NC - k] -> point k) is formed inside the first inverse stage.
            phs(WAVE_FFT, [&](int tid, Rg& r) {
                DS_SETPRIO(2);                                          // the serial part of a hop: ahead of other workgroups' wide phases
                if (LATE_PREFETCH && t + 1 < p.T) stage_hop_late(t + 1, old_half, tid, r);     // (that half: read by the first forward stage, idle since)
                if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, 2, 0, 1>(tid - INV_T0, NT, sh, nullptr, fb, 1, 0, 1);
                else if (WAVE_FFT && tid == NYQ_TID) {
                    cf Zn[M];
#pragma unroll
                    for (int m = 0; m < M; ++m) Zn[m] = mk(sh.zn[m], 0.0f);
#ifdef DS_ABLATE_NYQUIST
                    const cf Yn = Zn[0];
#else
                    const cf Yn = bin_program(sh.nyq, Zn, steer, NC, sh, p, frm_nyq, reset, spp_nyq, r.adn, r.apkn, ref_pow_row(t));
#endif
                    sh.Y[NC] = mk(Yn.x, 0.0f);                          // irfft ignores Im Y[N/2]
                }
            });
            phs(WAVE_FFT, [&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, false, 1, 2>(tid - INV_T0, NT, sh, fb, fa, 4, 0, 1); });
            phs(WAVE_FFT, [&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, false, 2, 0>(tid - INV_T0, NT, sh, fa, fb, 16, 0, 1); });
            if (NC == 128) {
                ex.stage([&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid - INV_T0, NT, sh, fb, fa, 64, 0, 1); });
            } else {
                ex.stage([&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 4, +1, false, 0, 0>(tid - INV_T0, NT, sh, fb, fa, 64, 0, 1); });
                if (NC == 512)
                    ex.stage([&](int tid, Rg&) { if (tid >= INV_T0) fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid - INV_T0, NT, sh, fa, fb, 256, 0, 1); });
            }
            const cf* Zi = INV_FINAL_IS_FA ? fa : fb;
            // ---- window, overlap-add, emit hop t ---------------------------------------------------
            ex.phase([&](int tid, Rg& r) {
                if constexpr (LATE_PREFETCH) ex.lds_load_wait();        // the next hop's samples have landed (issued five phases back); the phase's barrier publishes them
                if (tid < NC / 2) {
                    const int i = tid;
                    const float sc = 1.0f / (float)NC;
                    const float hn = WAVE_FFT ? 0.5f * sh.Y[NC].x : 0.0f;   // the Nyquist bin's share of every even (+) / odd (-) sample
                    cf z1 = Zi[i], z2 = Zi[i + NC / 2];
                    z1.x += hn; z1.y -= hn; z2.x += hn; z2.y -= hn;
                    const float y0 = sh.tb.win[2 * i] * (z1.x * sc), y1 = sh.tb.win[2 * i + 1] * (z1.y * sc);
                    const float o0 = (y0 + sh.tail[2 * i]) * p.out_scale, o1 = (y1 + sh.tail[2 * i + 1]) * p.out_scale;
                    const float t0 = sh.tb.win[HOP + 2 * i] * (z2.x * sc), t1 = sh.tb.win[HOP + 2 * i + 1] * (z2.y * sc);
                    sh.tail[2 * i] = t0;
                    sh.tail[2 * i + 1] = t1;
                    if constexpr (EARLY_STORE) {
                        if (t == T_run - 1) { tout[2 * i] = t0; tout[2 * i + 1] = t1; }      // the synthesis overlap the next call starts with
                    }
                    float* dst = p.y + yb + (long long)t * HOP + 2 * i;
#if defined(__HIP_DEVICE_COMPILE__)
                    typedef float f2_t __attribute__((ext_vector_type(2)));
                    f2_t o; o.x = o0; o.y = o1;
                    __builtin_nontemporal_store(o, reinterpret_cast<f2_t*>(dst));
#else
                    dst[0] = o0; dst[1] = o1;
#endif
                }
                if constexpr (EARLY_STORE) {
                    if (t == T_run - 1) {                               // the Nyquist bin's state (final since its pass, two barriers back) and the counters
                        store_nyquist_state(tid);
                        if (HAS_CNT && tid == 0) { cnt[0] = frm_cnt; cnt[1] = ell; cnt[2] = spp_cnt; }
                    }
                }
                if (!one_round) DS_SETPRIO(0);
            });
            old_half = new_half;
        }

        // ---- epilogue: state back to HBM (the chain tail; a call without hops) -----------------------
        if (!EARLY_STORE || T_run == 0)
        ex.phase([&](int tid, Rg& r) {
            {
                vec4* tin4 = reinterpret_cast<vec4*>(tin);
                for (int i = tid; i < M * HOP / 4; i += NT) {
                    const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                    store_state(&tin4[i], *reinterpret_cast<const vec4*>(&sh.xbuf[m][old_half * HOP + 4 * q]));
                }
                vec4* tout4 = reinterpret_cast<vec4*>(tout);
                for (int i = tid; i < HOP / 4; i += NT) store_state(&tout4[i], *reinterpret_cast<const vec4*>(&sh.tail[4 * i]));
            }
            if constexpr (ALGO == ALGO_AIC) {
                const AicState as(p, b, KP);
#pragma unroll
                for (int f = 0; f < SL::NF; ++f) as.st_once(f, tid, r.st[f]);
                if (tid < SL::NF) as.st(tid, NC, sh.nyq[tid]);
                if (p.aic_e != nullptr && tid < NC / 2) {               // overlap tails of the blocking-matrix synthesis
                    float* bt = p.aic_bmtail + (long long)b * M * HOP;
#pragma unroll
                    for (int c = 0; c < M; ++c) { bt[c * HOP + 2 * tid] = r.bmt[2 * c]; bt[c * HOP + 2 * tid + 1] = r.bmt[2 * c + 1]; }
                }
                if (p.T > 0) {                                          // the desired-signal frame the next call starts with
                    const float* last = p.aic_d + 2 * (((long long)b * p.T + (p.T - 1)) * K);
                    float* keep = p.aic_dprev + 2 * (long long)b * K;
                    keep[2 * tid] = last[2 * tid]; keep[2 * tid + 1] = last[2 * tid + 1];
                    if (tid == NYQ_TID) { keep[2 * NC] = last[2 * NC]; keep[2 * NC + 1] = last[2 * NC + 1]; }
                }
            } else {
                store_lane_state(tid, r);
                store_nyquist_state(tid);
            }
            if (HAS_CNT && tid == 0) { cnt[0] = frm_cnt; cnt[1] = ell; cnt[2] = spp_cnt; }
        });
    }
};

// ---------------------------------------------------------------------------------------------
// Stand-alone streaming STFT / ISTFT (Transform.stft / Transform.istft, transform/transform.py:430-481)
// for callers that drive the frame-level operators themselves.  Same FFT machinery as the fused
// engine; one workgroup per utterance.
//   STFT : x [B][..layout..]  ->  Y complex [B][T][K][M]   (p.y, p.y_batch_stride in floats)
//   ISTFT: Y complex [B][T][K][C] (p.x, p.x_batch_stride in floats) -> y [B][T*hop][C]   (C = p.method <= M)
// ---------------------------------------------------------------------------------------------
// OV = NFFT / hop: 2 (the half-overlap every beamformer object uses), or 4 for Transform(n_fft, hop_length = n_fft / 4)
// (transform.py:407-428 takes any hop; wpe.ipynb runs 75 % overlap): the LDS sample buffer becomes a ring of four quarter-frames.
// FRONT (with CDR; the SubbandGSC chain's front end as ONE kernel): the hop goes raw input -> DC notch -> FIR bank + channel mean -> analysis
// inside the workgroup.  The FIR windows (history + hop per channel) live in the second transform buffer and the taps in the first (both idle
// until the forward stages start), the carried history in the spectrum row `Y` this engine never uses: no LDS beyond the plain analysis.
template <int NFFT, int M, bool CDR = false, int OV = 2, bool FRONT = false> struct StftEngine {
    typedef Engine<NFFT, M, ALGO_FIXED, false> EB;
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2, NT = NC;
    typedef typename EB::Sh Sh;
    typedef typename EB::Rg Rg;
    static constexpr int HOPX = NFFT / OV, OVL = NFFT - HOPX;        // hop and carried overlap of this transform
    static_assert(!CDR || M >= 3, "McCDR takes microphones 0, 1 and 2");
    static_assert(OV == 2 || (OV == 4 && !CDR), "overlap");
    static_assert(!FRONT || (CDR && OV == 2 && HOP == NT), "the fused front end: one thread per hop sample");
    // can this shape keep the FIR history of an L-tap bank in LDS (the `Y` row) and its windows in the transform buffers?
    static constexpr bool front_fits(int L) {
        return L >= 1 && L <= 120 && (size_t)M * (L - 1) * sizeof(float) <= sizeof(Sh::Y) &&
               (size_t)M * ((((L - 1) + 3) & ~3) + 4 + HOP) * sizeof(float) <= sizeof(Sh::fb) && (size_t)M * ((L + 3) & ~3) * sizeof(float) <= sizeof(Sh::fa);
    }

    template <class Exec> static DS_HD void run(Exec& ex, const Params& p, int blk, Sh& sh) {
        const int b = p.batch0 + blk;
        const long long xb = (long long)blk * p.x_batch_stride;
        float* tin = p.tail_in + (long long)b * M * OVL;
        // FRONT: window row m = W[m * WL ...]: four zeros (the zero-padded taps read them), L - 1 history samples ending at OFF, the hop at OFF
        const int FL = FRONT ? p.fe_L : 1, FLp = (FL + 3) & ~3, OFF = (((FL - 1) + 3) & ~3) + 4, WL = OFF + HOP;
        float* const W = reinterpret_cast<float*>(&sh.fb[0][0]);
        float* const cs = reinterpret_cast<float*>(&sh.fa[0][0]);
        float* const hist = reinterpret_cast<float*>(&sh.Y[0]);
        cf* Yout = reinterpret_cast<cf*>(p.y + (long long)blk * p.y_batch_stride);
        int old_half = 0;
        constexpr int KP = plane_len(K);
        // McCDR's planes of this utterance (rows 0..8 of the McSpp stage's state), bin `tid` in registers for the call, the Nyquist bin on lane 0
        auto cdr_plane = [&](int f, int k) -> float& {                 // ds_ops.hpp st_index
            return p.cdr_st[(((long long)b * (aic_floats_per_bin(p.cdr_NF) >> 2) + (f >> 2)) * KP + k) * 4 + (f & 3)];
        };
        int frm = 0, ell = 0;
        if constexpr (CDR) { frm = p.cdr_cnt ? p.cdr_cnt[0] : p.cdr_frm; ell = p.cdr_cnt ? p.cdr_cnt[1] : p.cdr_ell; }
        const int fmin = (int)(500.0 * (2 * (K - 1)) / 16000.0), fmax = (int)(2000.0 * (2 * (K - 1)) / 16000.0);   // mcspp.py:258-259
        // band mean of 1 - Gamma of frame t (Gamma in sh.tail), summed in bin order by one lane like mcspp_qavg()
        auto band_mean = [&](int t) {
            float qsum = 0.0f;
            for (int j = fmin; j < fmax; ++j) qsum += 1.0f - sh.tail[j];
            p.cdr_qavg[(long long)b * p.T + t] = qsum / (float)(fmax - fmin);
        };
        ex.phase([&](int tid, Rg& r) {
            vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
            for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
            const vec4* tin4 = reinterpret_cast<const vec4*>(tin);
            for (int i = tid; i < M * OVL / 4; i += NT) {                      // OV = 4: the three carried quarters fill slots 0..2
                const int m = i / (OVL / 4), q = i - m * (OVL / 4);
                *reinterpret_cast<vec4*>(&sh.xbuf[m][4 * q]) = tin4[i];
            }
            if constexpr (OV == 2) {
                EB::prefetch_init(p, xb, tid, r);
                EB::prefetch(p, xb, 0, tid, r);
            }
            if constexpr (FRONT) {
                const float* cin = p.fe_cache_in + (long long)b * M * (FL - 1);
                for (int i = tid; i < M * (FL - 1); i += NT) hist[i] = cin[i];
                r.nm0 = r.nm1 = 0.0;
                if (tid < M) { r.nm0 = p.fe_mem[((long long)b * M + tid) * 2]; r.nm1 = p.fe_mem[((long long)b * M + tid) * 2 + 1]; }
            }
            if constexpr (CDR) {
#pragma unroll
                for (int f = 0; f < 9; ++f) r.cdr[f] = cdr_plane(f, tid);
                if (tid == 0) {
#pragma unroll
                    for (int f = 0; f < 9; ++f) r.cdrn[f] = cdr_plane(f, NC);
                }
            }
        });
        for (int t = 0; t < p.T; ++t) {
            const int new_half = old_half ^ 1;
            auto ph = [&](bool wave_local, auto f) { if (wave_local) ex.phase_wave(f); else ex.phase(f); };   // see Engine::run
            cf* fa = &sh.fa[0][0];
            cf* fb = &sh.fb[0][0];
            if constexpr (OV == 4) {
                const int slot = (t + 3) & 3;                                  // the new quarter replaces the oldest one
                ex.phase([&](int tid, Rg&) {
                    for (int i = tid; i < M * HOPX; i += NT) {
                        int m, n;
                        long long off;
                        if (p.x_sample_stride == 1) { m = i / HOPX; n = i - m * HOPX; off = (long long)m * p.x_chan_stride + (long long)t * HOPX + n; }
                        else { n = i / M; m = i - n * M; off = ((long long)t * HOPX + n) * M + m; }
                        sh.xbuf[m][slot * HOPX + n] = p.x[xb + off];
                    }
                });
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, 3, 0, 1>(tid, NT, sh, nullptr, fa, 1, (slot + 1) & 3, M); });
            } else if constexpr (FRONT) {
                // ---- raw hop, history and taps into the windows (x is channel-major here: x_sample_stride == 1) --------------------------
                ex.phase([&](int tid, Rg& r) {
#pragma unroll
                    for (int i = 0; i < EB::NPRE; ++i) {
                        const int v = tid + i * NT;
                        if (v < EB::NV4) { const int m = v / (HOP / 4), q = v - m * (HOP / 4); *reinterpret_cast<vec4*>(&W[m * WL + OFF + 4 * q]) = r.pre[i]; }
                    }
                    for (int m = 0; m < M; ++m) {
                        for (int h = tid; h < FL - 1; h += NT) W[m * WL + OFF - (FL - 1) + h] = hist[m * (FL - 1) + h];
                        if (tid < 4) W[m * WL + OFF - (FL - 1) - 4 + tid] = 0.0f;
                        for (int j = tid; j < FLp; j += NT) cs[m * FLp + j] = j < FL ? p.fe_coef[(long long)j * M + m] : 0.0f;
                    }
                    if (t + 1 < p.T) EB::prefetch(p, xb, t + 1, tid, r);
                    if (t > 0 && tid == NT - 1) band_mean(t - 1);
                });
                // ---- DC notch, in place: one lane per channel, serial in time (ds_ops.hpp td_dcnotch: same statements) ----------------------
                ex.phase([&](int tid, Rg& r) {
                    if (tid >= M) return;
                    const double rr = p.fe_radius, r2 = rr * rr + 0.7 * (1.0 - rr) * (1.0 - rr);   // notch_den2 / notch_step of ds_ops.hpp, word for word
                    float* row = W + tid * WL + OFF;
                    double m0 = r.nm0, m1 = r.nm1;
                    for (int i = 0; i < HOP; ++i) {
                        const double vin = (double)row[i];
                        const double vout = m0 + vin;
                        m0 = m1 + 2.0 * (-vin + rr * vout);
                        m1 = vin - r2 * vout;
                        row[i] = (float)(rr * vout);
                    }
                    r.nm0 = m0; r.nm1 = m1;
                });
                // ---- FIR bank: thread o = output sample o of every channel, taps in ascending order (td_fir); channel mean; new history -----
                ex.phase([&](int tid, Rg&) {
                    float mean = 0.0f;
                    for (int m = 0; m < M; ++m) {
                        const float* w = W + m * WL + OFF + tid;
                        const float* c = cs + m * FLp;
                        float acc = 0.0f;
                        for (int jb = 0; jb < FLp; jb += 4) {
                            const vec4 c4 = *reinterpret_cast<const vec4*>(c + jb);
                            acc = fma_(c4.x, w[-jb], acc);
                            acc = fma_(c4.y, w[-jb - 1], acc);
                            acc = fma_(c4.z, w[-jb - 2], acc);
                            acc = fma_(c4.w, w[-jb - 3], acc);
                        }
                        sh.xbuf[m][new_half * HOP + tid] = acc;
                        mean += acc;
                        for (int h = tid; h < FL - 1; h += NT) hist[m * (FL - 1) + h] = W[m * WL + OFF + HOP - (FL - 1) + h];
                    }
                    p.fe_fixed[((long long)b * p.T + t) * HOP + tid] = mean / (float)M;
                });
                ph(EB::WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, true, 0, 1>(tid, NT, sh, nullptr, fa, 1, old_half, M); });
            } else {
            ph(EB::WAVE_FFT && p.x_sample_stride == 1, [&](int tid, Rg& r) {
                EB::commit(p, sh, new_half, tid, r);
                if (t + 1 < p.T) EB::prefetch(p, xb, t + 1, tid, r);
                if constexpr (CDR) { if (t > 0 && tid == NT - 1) band_mean(t - 1); }      // Gamma of the previous frame is behind a barrier by now
            });
            ph(EB::WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, true, 0, 1>(tid, NT, sh, nullptr, fa, 1, old_half, M); });
            }
            ph(EB::WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 1, 2>(tid, NT, sh, fa, fb, 4, 0, M); });
            ph(EB::WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 2, 0>(tid, NT, sh, fb, fa, 16, 0, M); });
            if (NC == 128) {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, M); });
            } else {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 4, -1, false, 0, 0>(tid, NT, sh, fa, fb, 64, 0, M); });
                if (NC == 512)
                    ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, -1, false, 0, 0>(tid, NT, sh, fb, fa, 256, 0, M); });
            }
            const cf* F = EB::FWD_FINAL_IS_FB ? fb : fa;
            ex.phase([&](int tid, Rg& r) {
                const int k = tid, k2 = (NC - k) & (NC - 1);
                const cf w = sh.tb.tw[k];
                cf* dst = Yout + ((long long)t * K + k) * M;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const cf A = F[m * Sh::NCP + k], B = F[m * Sh::NCP + k2];
                    const cf E = cscale(cadd_c(A, B), 0.5f);
                    const cf D = csub_c(A, B);
                    const cf O = cdiv_2j(D);
                    cf Z = cfma(E, w, O);
                    if (k == 0) Z.y = 0.0f;
                    dst[m] = Z;
                    if constexpr (CDR) { if (m < 3) r.Z[m] = Z; }
                }
                if (k == 0) {                                  // Nyquist bin
                    cf* dn = Yout + ((long long)t * K + NC) * M;
#pragma unroll
                    for (int m = 0; m < M; ++m) {
                        const cf F0 = F[m * Sh::NCP];
                        dn[m] = mk(F0.x - F0.y, 0.0f);
                        if constexpr (CDR) { if (m < 3) sh.zn[m] = F0.x - F0.y; }
                    }
                    if constexpr (CDR) sh.pw[NC] = sh.zn[0] * sh.zn[0];
                }
                if constexpr (CDR) sh.pw[k] = cabs2(r.Z[0]);
            });
            if constexpr (CDR) {
                // ---- McCDR of frame t: Gamma to HBM (McSpp's input) and to LDS (the band mean) ----------------------------------------
                const bool reset = (frm != 0) && (ell % p.cdr_L == 0);
                ex.phase([&](int tid, Rg& r) {
                    const int k = tid;
                    cf x12 = mk(r.cdr[2], r.cdr[3]);
                    const float g = mccdr_frame(r.cdr[0], r.cdr[1], x12, r.cdr + 4, r.Z[1], r.Z[2], p.cdr_fn[k], k > 0 ? sh.pw[k - 1] : 0.0f,
                                                sh.pw[k], sh.pw[k + 1], k, K, frm, reset, p.cdr_L);
                    r.cdr[2] = x12.x; r.cdr[3] = x12.y;
                    float* gout = p.cdr_gamma + ((long long)b * p.T + t) * K;
                    gout[k] = g;
                    sh.tail[k] = g;
                    if (k == 0) {                              // ... and of the Nyquist bin
                        cf xn = mk(r.cdrn[2], r.cdrn[3]);
                        gout[NC] = mccdr_frame(r.cdrn[0], r.cdrn[1], xn, r.cdrn + 4, mk(sh.zn[1], 0.0f), mk(sh.zn[2], 0.0f), p.cdr_fn[NC],
                                               sh.pw[NC - 1], sh.pw[NC], 0.0f, NC, K, frm, reset, p.cdr_L);
                        r.cdrn[2] = xn.x; r.cdrn[3] = xn.y;
                    }
                });
                if (reset) ell = 0;
                frm += 1; ell += 1;
            }
            old_half = new_half;
        }
        ex.phase([&](int tid, Rg& r) {
            vec4* tin4 = reinterpret_cast<vec4*>(tin);
            if constexpr (OV == 4) {                                           // the three newest quarters, oldest first
                const int first = p.T & 3;                                     // = (slot of the last hop + 2) & 3; 0 for an empty call
                for (int i = tid; i < M * OVL / 4; i += NT) {
                    const int m = i / (OVL / 4), q = i - m * (OVL / 4), qq = q / (HOPX / 4), e = q - qq * (HOPX / 4);
                    tin4[i] = *reinterpret_cast<const vec4*>(&sh.xbuf[m][((first + qq) & 3) * HOPX + 4 * e]);
                }
            } else {
            for (int i = tid; i < M * HOP / 4; i += NT) {
                const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                tin4[i] = *reinterpret_cast<const vec4*>(&sh.xbuf[m][old_half * HOP + 4 * q]);
            }
            }
            if constexpr (FRONT) {
                float* cout = p.fe_cache_out + (long long)b * M * (FL - 1);
                for (int i = tid; i < M * (FL - 1); i += NT) cout[i] = hist[i];
                if (tid < M) { p.fe_mem[((long long)b * M + tid) * 2] = r.nm0; p.fe_mem[((long long)b * M + tid) * 2 + 1] = r.nm1; }
            }
            if constexpr (CDR) {
                if (p.T > 0 && tid == NT - 1) band_mean(p.T - 1);
#pragma unroll
                for (int f = 0; f < 9; ++f) cdr_plane(f, tid) = r.cdr[f];
                if (tid == 0) {
#pragma unroll
                    for (int f = 0; f < 9; ++f) cdr_plane(f, NC) = r.cdrn[f];
                }
            }
        });
    }
};

template <int NFFT, int M, int OV = 2> struct SharedIstft {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = OV == 2 ? NFFT / 2 : NFFT;   // OV = 4: a ring of four quarter-frames
    static constexpr int NCP = NC + NC / 4;
    cf fa[M][NCP];
    cf fb[M][NCP];
    alignas(16) Tables<NFFT> tb;
    alignas(16) float tail[M][HOP];
};

// OV = 4 (hop = NFFT / 4): the overlap-add accumulator is a ring of four quarter-frames in LDS; a frame adds its r-th quarter to slot
// (head + r) & 3, the head slot leaves as this hop's output and is cleared for the frame after next's last quarter.  Between calls the
// three carried quarters (Transform.previous_output, transform.py:476-477) are kept oldest first, so a chunked run repeats the
// single-call arithmetic step for step.
template <int NFFT, int M, int OV = 2> struct IstftEngine {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2, NT = NC, HOPX = NFFT / OV, OVL = NFFT - HOPX;
    typedef SharedIstft<NFFT, M, OV> Sh;
    struct Rg { int unused; };

    template <class Exec> static DS_HD void run(Exec& ex, const Params& p, int blk, Sh& sh) {
        const int b = p.batch0 + blk;
        const int C = p.method;                                // channels in this call (<= M)
        const cf* Yin = reinterpret_cast<const cf*>(p.x + (long long)blk * p.x_batch_stride);
        float* yout = p.y + (long long)blk * p.y_batch_stride;
        float* tout = p.tail_out + (long long)b * M * OVL;
        ex.phase([&](int tid, Rg&) {
            vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
            for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
            const vec4* t4 = reinterpret_cast<const vec4*>(tout);
            for (int i = tid; i < M * OVL / 4; i += NT) {
                const int m = i / (OVL / 4), q = i - m * (OVL / 4);
                *reinterpret_cast<vec4*>(&sh.tail[m][4 * q]) = t4[i];
            }
            if constexpr (OV == 4) {
                for (int i = tid; i < M * HOPX; i += NT) sh.tail[i / HOPX][OVL + i % HOPX] = 0.0f;     // slot 3: nothing accumulated yet
            }
        });
        cf* fa = &sh.fa[0][0];
        cf* fb = &sh.fb[0][0];
        for (int t = 0; t < p.T; ++t) {
            ex.phase([&](int tid, Rg&) {
                const int k = tid;
                const cf* Yt = Yin + (long long)t * K * C;
                const cf w = cconj(sh.tb.tw[k]);
                for (int c = 0; c < C; ++c) {
                    cf A = Yt[(long long)k * C + c], B = Yt[(long long)(NC - k) * C + c];
                    if (k == 0) { A.y = 0.0f; B.y = 0.0f; }     // irfft ignores Im Y[0], Im Y[N/2]
                    const cf E = cscale(cadd_c(A, B), 0.5f);
                    const cf O = cmul(cscale(csub_c(A, B), 0.5f), w);
                    fa[c * Sh::NCP + k] = cadd_jd<+1>(E, O);
                }
            });
            auto ph = [&](bool wave_local, auto f) { if (wave_local) ex.phase_wave(f); else ex.phase(f); };   // see Engine::run
            constexpr bool WAVE_FFT = NC == 256;
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 0, 1>(tid, NT, sh, fa, fb, 1, 0, C); });
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 1, 2>(tid, NT, sh, fb, fa, 4, 0, C); });
            ph(WAVE_FFT, [&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 2, 0>(tid, NT, sh, fa, fb, 16, 0, C); });
            if (NC == 128) {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid, NT, sh, fb, fa, 64, 0, C); });
            } else {
                ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 4, +1, false, 0, 0>(tid, NT, sh, fb, fa, 64, 0, C); });
                if (NC == 512)
                    ex.phase([&](int tid, Rg&) { fft_stage<NFFT, M, 2, +1, false, 0, 0>(tid, NT, sh, fa, fb, 256, 0, C); });
            }
            const cf* Zi = (NC != 512) ? fa : fb;
            if constexpr (OV == 4) {
                const int head = t & 3;
                ex.phase([&](int tid, Rg&) {
                    const int i = tid, qr = (2 * i) / HOPX, e = 2 * i - qr * HOPX, at = ((head + qr) & 3) * HOPX + e;
                    const float sc = 1.0f / (float)NC;
                    for (int c = 0; c < C; ++c) {
                        const cf z = Zi[c * Sh::NCP + i];
                        const float a0 = sh.tail[c][at] + sh.tb.win[2 * i] * (z.x * sc), a1 = sh.tail[c][at + 1] + sh.tb.win[2 * i + 1] * (z.y * sc);
                        if (qr == 0) {
                            yout[((long long)t * HOPX + e) * C + c] = a0 * p.out_scale;
                            yout[((long long)t * HOPX + e + 1) * C + c] = a1 * p.out_scale;
                            sh.tail[c][at] = 0.0f; sh.tail[c][at + 1] = 0.0f;
                        } else {
                            sh.tail[c][at] = a0; sh.tail[c][at + 1] = a1;
                        }
                    }
                });
            } else
            ex.phase([&](int tid, Rg&) {
                if (tid < NC / 2) {
                    const int i = tid;
                    const float sc = 1.0f / (float)NC;
                    for (int c = 0; c < C; ++c) {
                        const cf z1 = Zi[c * Sh::NCP + i], z2 = Zi[c * Sh::NCP + i + NC / 2];
                        const float y0 = sh.tb.win[2 * i] * (z1.x * sc), y1 = sh.tb.win[2 * i + 1] * (z1.y * sc);
                        const float o0 = (y0 + sh.tail[c][2 * i]) * p.out_scale, o1 = (y1 + sh.tail[c][2 * i + 1]) * p.out_scale;
                        sh.tail[c][2 * i] = sh.tb.win[HOP + 2 * i] * (z2.x * sc);
                        sh.tail[c][2 * i + 1] = sh.tb.win[HOP + 2 * i + 1] * (z2.y * sc);
                        yout[((long long)t * HOP + 2 * i) * C + c] = o0;
                        yout[((long long)t * HOP + 2 * i + 1) * C + c] = o1;
                    }
                }
            });
        }
        ex.phase([&](int tid, Rg&) {
            vec4* t4 = reinterpret_cast<vec4*>(tout);
            if constexpr (OV == 4) {
                const int head = p.T & 3;                                     // the slot the next frame's first quarter goes to
                for (int i = tid; i < M * OVL / 4; i += NT) {
                    const int m = i / (OVL / 4), q = i - m * (OVL / 4), qq = q / (HOPX / 4), e = q - qq * (HOPX / 4);
                    t4[i] = *reinterpret_cast<const vec4*>(&sh.tail[m][((head + qq) & 3) * HOPX + 4 * e]);
                }
            } else
            for (int i = tid; i < M * HOP / 4; i += NT) {
                const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                t4[i] = *reinterpret_cast<const vec4*>(&sh.tail[m][4 * q]);
            }
        });
    }
};

// ---------------------------------------------------------------------------------------------
// Single-channel transforms, one row (utterance) per WAVEFRONT: the 64 lanes of a wave run every phase of their row's transform
// (2 or 4 points per lane in the split / merge phases, NC / 256 butterflies per lane in the FFT stages), so nothing inside the
// hop loop needs a workgroup barrier and no lane idles while "its" channel is in another wave's hands.  Four rows per workgroup
// share the tables.  Same per-element arithmetic as StftEngine<NFFT, 1> / IstftEngine<NFFT, 1>.
//   STFT : x rows (p.x + row * x_batch_stride, contiguous samples)  ->  Y complex [row][T][K]  (p.y + row * y_batch_stride floats)
//   ISTFT: Y complex [row][T][K] (p.x + row * x_batch_stride floats) -> y rows (p.y + row * y_batch_stride)
// ---------------------------------------------------------------------------------------------
template <int NFFT> struct SharedRows {
    static constexpr int N = NFFT, NC = NFFT / 2, HOP = NFFT / 2, NCP = NC + NC / 4, ROWS = 4;
    alignas(16) Tables<NFFT> tb;
    alignas(16) float xrow[ROWS][N];      // STFT: [old hop | new hop] per row
    cf fa[ROWS][NCP];
    cf fb[ROWS][NCP];
    alignas(16) float tail[ROWS][HOP];    // ISTFT: overlap-add tail per row
};
template <int NFFT> struct RowView {         // what fft_stage sees of one row
    static constexpr int NCP = NFFT / 2 + NFFT / 8;
    const Tables<NFFT>& tb;
    float (*xbuf)[NFFT];
};
template <int NV> struct RowRegs { vec4 pre[NV]; };     // (sized to the shape: with two slots for the one a 512-point row uses the struct stayed in scratch)

template <int NFFT> struct StftRowsEngine {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2, ROWS = 4, NT = 64 * ROWS, NV = HOP / 256;
    static_assert(HOP % 256 == 0, "a lane carries whole 16-byte pieces of a hop");
    typedef SharedRows<NFFT> Sh;
    typedef RowRegs<NV> Rg;

    template <class Exec> static DS_HD void run(Exec& ex, const Params& p, int blk, Sh& sh) {
        auto where = [&](int tid, int& w, int& lane, int& row) { w = tid >> 6; lane = tid & 63; row = blk * ROWS + w; return row < p.rows; };
        ex.phase([&](int tid, Rg& r) {
            vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
            for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
            int w, lane, row;
            if (!where(tid, w, lane, row)) return;
            const vec4* tin4 = reinterpret_cast<const vec4*>(p.tail_in + (long long)row * HOP);
            const vec4* x4 = reinterpret_cast<const vec4*>(p.x + (long long)row * p.x_batch_stride);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                *reinterpret_cast<vec4*>(&sh.xrow[w][4 * (lane + 64 * i)]) = tin4[lane + 64 * i];
                r.pre[i] = x4[lane + 64 * i];
            }
        });
        int old_half = 0;
        for (int t = 0; t < p.T; ++t) {
            const int new_half = old_half ^ 1;
            ex.phase_wave([&](int tid, Rg& r) {
                int w, lane, row;
                if (!where(tid, w, lane, row)) return;
                const vec4* x4 = reinterpret_cast<const vec4*>(p.x + (long long)row * p.x_batch_stride + (long long)(t + 1) * HOP);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    *reinterpret_cast<vec4*>(&sh.xrow[w][new_half * HOP + 4 * (lane + 64 * i)]) = r.pre[i];
                    if (t + 1 < p.T) r.pre[i] = x4[lane + 64 * i];
                }
            });
            auto stage = [&](auto f) {
                ex.phase_wave([&](int tid, Rg&) {
                    int w, lane, row;
                    if (!where(tid, w, lane, row)) return;
                    RowView<NFFT> rv = {sh.tb, &sh.xrow[w]};
                    f(lane, rv, &sh.fa[w][0], &sh.fb[w][0]);
                });
            };
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, -1, true, 0, 1>(l, 64, rv, nullptr, fa, 1, old_half, 1); });
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, -1, false, 1, 2>(l, 64, rv, fa, fb, 4, 0, 1); });
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, -1, false, 2, 0>(l, 64, rv, fb, fa, 16, 0, 1); });
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, -1, false, 0, 0>(l, 64, rv, fa, fb, 64, 0, 1); });
            if (NC == 512)
                stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 2, -1, false, 0, 0>(l, 64, rv, fb, fa, 256, 0, 1); });
            ex.phase_wave([&](int tid, Rg&) {
                int w, lane, row;
                if (!where(tid, w, lane, row)) return;
                const cf* F = NC == 512 ? &sh.fa[w][0] : &sh.fb[w][0];
                cf* Yt = reinterpret_cast<cf*>(p.y + (long long)row * p.y_batch_stride) + (long long)t * K;
                for (int k = lane; k < NC; k += 64) {
                    const int k2 = (NC - k) & (NC - 1);
                    const cf wk = sh.tb.tw[k];
                    const cf A = F[k], B = F[k2];
                    const cf E = cscale(cadd_c(A, B), 0.5f);
                    const cf D = csub_c(A, B);
                    const cf O = cdiv_2j(D);
                    cf Z = cfma(E, wk, O);
                    if (k == 0) Z.y = 0.0f;
                    Yt[k] = Z;
                }
                if (lane == 0) { const cf F0 = F[0]; Yt[NC] = mk(F0.x - F0.y, 0.0f); }      // Nyquist bin
            });
            old_half = new_half;
        }
        ex.phase([&](int tid, Rg&) {
            int w, lane, row;
            if (!where(tid, w, lane, row)) return;
            vec4* tin4 = reinterpret_cast<vec4*>(p.tail_in + (long long)row * HOP);
#pragma unroll
            for (int i = 0; i < NV; ++i) tin4[lane + 64 * i] = *reinterpret_cast<const vec4*>(&sh.xrow[w][old_half * HOP + 4 * (lane + 64 * i)]);
        });
    }
};

template <int NFFT> struct IstftRowsEngine {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2, ROWS = 4, NT = 64 * ROWS, NV = HOP / 256, NPL = NC / 64;
    typedef SharedRows<NFFT> Sh;
    struct Rg { cf a[NPL], b[NPL]; };        // the next frame's Y[k] and Y[NC - k] of this lane's bins, in flight behind the current frame

    template <class Exec> static DS_HD void run(Exec& ex, const Params& p, int blk, Sh& sh) {
        auto where = [&](int tid, int& w, int& lane, int& row) { w = tid >> 6; lane = tid & 63; row = blk * ROWS + w; return row < p.rows; };
        ex.phase([&](int tid, Rg&) {
            vec4* tb4 = reinterpret_cast<vec4*>(&sh.tb);
            for (int i = tid; i < Tables<NFFT>::NV4; i += NT) tb4[i] = p.tables[i];
            int w, lane, row;
            if (!where(tid, w, lane, row)) return;
            const vec4* t4 = reinterpret_cast<const vec4*>(p.tail_out + (long long)row * HOP);
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<vec4*>(&sh.tail[w][4 * (lane + 64 * i)]) = t4[lane + 64 * i];
        });
        auto fetch = [&](int row, int lane, int t, Rg& r) {
            const cf* Yt = reinterpret_cast<const cf*>(p.x + (long long)row * p.x_batch_stride) + (long long)t * K;
#pragma unroll
            for (int j = 0; j < NPL; ++j) { const int k = lane + 64 * j; r.a[j] = Yt[k]; r.b[j] = Yt[NC - k]; }
        };
        ex.phase_wave([&](int tid, Rg& r) {
            int w, lane, row;
            if (where(tid, w, lane, row) && p.T > 0) fetch(row, lane, 0, r);
        });
        for (int t = 0; t < p.T; ++t) {
            ex.phase_wave([&](int tid, Rg& r) {
                int w, lane, row;
                if (!where(tid, w, lane, row)) return;
                cf* fa = &sh.fa[w][0];
#pragma unroll
                for (int j = 0; j < NPL; ++j) {
                    const int k = lane + 64 * j;
                    const cf wk = cconj(sh.tb.tw[k]);
                    cf A = r.a[j], B = r.b[j];
                    if (k == 0) { A.y = 0.0f; B.y = 0.0f; }     // irfft ignores Im Y[0], Im Y[N/2]
                    const cf E = cscale(cadd_c(A, B), 0.5f);
                    const cf O = cmul(cscale(csub_c(A, B), 0.5f), wk);
                    fa[k] = cadd_jd<+1>(E, O);
                }
                if (t + 1 < p.T) fetch(row, lane, t + 1, r);
            });
            auto stage = [&](auto f) {
                ex.phase_wave([&](int tid, Rg&) {
                    int w, lane, row;
                    if (!where(tid, w, lane, row)) return;
                    RowView<NFFT> rv = {sh.tb, &sh.xrow[w]};
                    f(lane, rv, &sh.fa[w][0], &sh.fb[w][0]);
                });
            };
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, +1, false, 0, 1>(l, 64, rv, fa, fb, 1, 0, 1); });
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, +1, false, 1, 2>(l, 64, rv, fb, fa, 4, 0, 1); });
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, +1, false, 2, 0>(l, 64, rv, fa, fb, 16, 0, 1); });
            stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 4, +1, false, 0, 0>(l, 64, rv, fb, fa, 64, 0, 1); });
            if (NC == 512)
                stage([&](int l, RowView<NFFT>& rv, cf* fa, cf* fb) { fft_stage<NFFT, 1, 2, +1, false, 0, 0>(l, 64, rv, fa, fb, 256, 0, 1); });
            ex.phase_wave([&](int tid, Rg&) {
                int w, lane, row;
                if (!where(tid, w, lane, row)) return;
                const cf* Zi = NC == 512 ? &sh.fb[w][0] : &sh.fa[w][0];
                float* yrow = p.y + (long long)row * p.y_batch_stride + (long long)t * HOP;
                const float sc = 1.0f / (float)NC;
                for (int i = lane; i < NC / 2; i += 64) {
                    const cf z1 = Zi[i], z2 = Zi[i + NC / 2];
                    const float y0 = sh.tb.win[2 * i] * (z1.x * sc), y1 = sh.tb.win[2 * i + 1] * (z1.y * sc);
                    const float o0 = (y0 + sh.tail[w][2 * i]) * p.out_scale, o1 = (y1 + sh.tail[w][2 * i + 1]) * p.out_scale;
                    sh.tail[w][2 * i] = sh.tb.win[HOP + 2 * i] * (z2.x * sc);
                    sh.tail[w][2 * i + 1] = sh.tb.win[HOP + 2 * i + 1] * (z2.y * sc);
                    yrow[2 * i] = o0; yrow[2 * i + 1] = o1;
                }
            });
        }
        ex.phase([&](int tid, Rg&) {
            int w, lane, row;
            if (!where(tid, w, lane, row)) return;
            vec4* t4 = reinterpret_cast<vec4*>(p.tail_out + (long long)row * HOP);
#pragma unroll
            for (int i = 0; i < NV; ++i) t4[lane + 64 * i] = *reinterpret_cast<const vec4*>(&sh.tail[w][4 * (lane + 64 * i)]);
        });
    }
};

}  // namespace ds
