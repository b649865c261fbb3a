// ds_ops.hpp — frame-level operators of the hot path as stand-alone (utterance, bin) programs:
// the L2 objects the reference lets callers drive frame by frame:
//   NoiseEstimationMCRA.estimation   noise_estimation/mcra.py:27-77
//   McMcra.estimation                noise_estimation/mc_mcra.py:179-224
//   NsOmlsaMulti.estimation          noise_estimation/omlsa_multi.py:73-156
//   SubbandLMS / SubbandLmsMc.update adaptivefilter/SubbandLMS.py:28-84, SubbandLmsMc.py:144-191
//   SubbandRLS.update                adaptivefilter/SubbandRLS.py:44-71
// One GPU thread owns one (utterance, bin) and walks the T frames of the call; state is a
// structure-of-arrays [B][NF][KP] of floats so that lane k touches address k: coalesced.
// Written single-source (DS_HD) so tests/emul can run the same code serially on the CPU.
#pragma once
#include "ds_core.hpp"
#include "ds_linalg64.hpp"
#include "ds_wpe.hpp"     // OP_WPE handles run the lane-parallel block program of ds_wpe.hpp, not a per-thread operator

namespace ds {

enum { OP_MCRA = 0, OP_MCMCRA = 1, OP_OMLSA = 2, OP_SUBLMS = 3, OP_SUBRLS = 4, OP_MCSPPBASE = 5, OP_WPE = 6, OP_MCCDR = 7, OP_MCSPP = 8, OP_STEERING = 9,
       OP_MVDRW = 10, OP_ADAPTIVE = 11, OP_MCSPP_LEAN = 12,      // LEAN: McSpp without the MVDR / matrix outputs (the SubbandGSC chain)
       OP_MCSPP_STEADY = 13,                                     // ... and its variant for calls from frame 5 on without PMWF weights
       OP_PMWFW = 14, OP_GEV = 15, OP_BAN = 16, OP_PHASECORR = 17,
       OP_MCSPP_STEADY_FAN = 18 };  // OP_MCSPP_STEADY with the SubbandGSC chain's M RLS blocking filters (op_subrls_fan) in the same thread: both read the
                                    // same frame of the same bin, so the spectra are fetched once and the filters' streaming traffic hides behind McSpp's arithmetic   // the free functions of beamformer/beamformer.py:34-130 (mvdr.ipynb's GEV flow)

struct OpParams {
    int B, K, KP, T;          // utterances, bins, padded plane length, frames in this call
    float* st;                // state [B][NF][KP]
    int NF;
    const float* in0;         // op-specific inputs (device)
    const float* in1;
    const float* in2;
    float* out0;              // op-specific outputs (device)
    float* out1;
    float* out2;
    float* out3;              // optional extra outputs (McSpp: Phi_xx, Phi_vv_inv as they stand at the end of estimation())
    float* out4;
    int M;                    // channels: mics (McMcra), beam + references (OMLSA), filter channels (subband LMS)
    int N;                    // filter taps
    int frm_cnt, ell, L;      // MCRA counters before the first frame of the call (uniform over the batch)
    int first_frame;          // OMLSA: 1 until the first estimation() has run
    int in_complex;           // MCRA: input is complex, take |.|^2 (mcra.py:29-30)
    int has_p;                // subband LMS: per-bin update probability given
    int norm;                 // subband LMS: power normalisation on (SubbandAF.py:18)
    float mu, alpha, reg, lam;   // step, power smoothing, regulariser (update(alpha=1e-4)), RLS forgetting factor
    int x_fan;                // subband LMS / RLS: instances b share the reference input (and p) of utterance b / x_fan (SubbandGSC's M
                              // blocking filters as one batch of B * M single-channel filters); 1 = one input per instance
    int p_complement;         // subband LMS: use 1 - p (SubbandGSC.py:232 passes p = 1 - p to the canceller)
    int d_interleaved;        // subband LMS / RLS with x_fan > 1: the desired signals of the x_fan instances of an utterance are the channels of
                              // one multichannel spectrum, in1 = complex [B / x_fan][T][K][x_fan] (SubbandGSC: the aligned-channel STFT)
    float* d_prev;            // subband LMS: desired signal delayed by one frame (SubbandGSC.py:226 delay_fbf): frame t takes in1[t - 1], frame 0
                              // takes d_prev complex [B][K], which the operator then replaces by the last frame of in1; null = no delay
    const cf* steer;          // OP_ADAPTIVE: steering vector a [K][M] (or [B][K][M] with steer_batch_stride)
    long long steer_batch_stride;
    int method;               // OP_ADAPTIVE: METHOD_SRC / DS / MVDR
    float alpha_v, gate, diag;   // OP_ADAPTIVE: adaptivebeamformer.py:66,94,89
    float diag_floor;         // OP_ADAPTIVE: pivot_floor(diag), see Params::diag_floor
    float beta_v;             // OP_ADAPTIVE: 1 - alpha_v (complement_of)
    const int* dev_cnt;       // optional device-resident {frm_cnt, ell, first_frame}: when set, the kernel takes the uniform counters from there instead
                              // of from this struct, so that a captured hipGraph of the launch stays valid as the stream advances (ds_tick_kernel
                              // advances them behind every launch)
    int in2_b0;               // McSpp: in2 (the band average per (utterance, frame)) starts at this utterance (the kernel's own LDS copy)
    TickArgs tick;            // counters of an EARLIER stage of the chain to advance (thread 0 of block 0; cnt null = none)
    int repeat;               // McSpp: estimation(repeat=True), a second estimation_core after the noise update (mcspp.py:280-282)
    // OP_MCSPP_STEADY_FAN: the blocking filters' side of the fused operator (the McSpp side uses the fields above as OP_MCSPP_STEADY does)
    float* fan_st;            // state of the B * M SubbandRLS instances (the DS_ALGO_SUBRLS stage's planes, fan form: see op_subrls_fan)
    int fan_NF;
    const float* fan_x;       // their shared reference input, complex [B][T][K] (the fixed beamformer's spectrum)
    float* fan_e;             // their error spectra, complex [B * M][T][K]
    float fan_lam, fan_mu;
    const void* fan_ctx;      // set by the kernel: the OpCtx of fan_st (device) / the OpParams copy (CPU emulator)
    float* spill;             // lean McSpp at 6 microphones: per-lane parking space (LDS on the GPU) for Phi_vv while the solves run; element f of
    int spill_stride;         // this lane at spill[f * spill_stride]; null = keep everything in registers
};

// State of the per-thread operators in memory: float f of bin k of utterance b at [b][f / 4][k][f % 4] — planes of float4, so that a lane
// reads and writes its bin's state as 16-byte accesses (the frame kernels' planes reach 6.2 TB/s that way, 4-byte planes 4.85:
// scratch/micro/planes_bw.hip); NF is rounded up to a multiple of 4 per utterance.  (Round 2 kept [b][f][k] planes of single floats.)
DS_HD constexpr int st_floats_per_bin(int NF) { return (NF + 3) & ~3; }
DS_HD long long st_index(int b, int f, int k, int NF, int KP) {
#ifdef DS_OLD_OPSTATE      // bisection builds only: round 2's [b][f][k] planes (the chain tail / CDR kernels keep the new layout: do not mix)
    return ((long long)b * st_floats_per_bin(NF) + f) * KP + k;
#else
    return (((long long)b * (st_floats_per_bin(NF) >> 2) + (f >> 2)) * KP + k) * 4 + (f & 3);
#endif
}

// Per-bin state access st_at(p, b, plane, k).  On the GPU it is a buffer access: the descriptor starts at the state of the first utterance
// of the workgroup (wave-uniform), the lane contributes ONE 32-bit offset and the plane offset travels in an SGPR — one address register
// per lane instead of a 64-bit pointer per plane kept live from the loads to the stores (150 registers for the 75 planes of a 6-mic McSpp).
#if defined(__HIP_DEVICE_COMPILE__)
struct OpCtx : OpParams {
    __amdgpu_buffer_rsrc_t rs;    // state from utterance b0 on
    int b0;
};
// cache policy of the operators' state accesses: 2 = nt.  An operator reads every state line once and writes it once per launch and the
// next launch (possibly on another XCD) is the next to touch it: streamed, like the frame kernels' load_state / store_state.  It pays since
// the plane rows are whole 128-byte lines (plane_len(): cfg5 +8.9 %, cfg4 +2.3 % at one hop per call, nothing lost at 10 s per call;
// on the 260-lane rows of rounds 1 - 3 it had gained nothing: profiles/r04a/opstate_nt_lines_ab.txt, wpe_nt_ab.txt).  -DDS_OPSTATE_POLICY=0: A/B
#if !defined(DS_OPSTATE_POLICY)
#if defined(DS_PLAIN_STATE)
#define DS_OPSTATE_POLICY 0
#else
#define DS_OPSTATE_POLICY 2
#endif
#endif
#if !defined(DS_OPSTATE_STORE_POLICY)
#define DS_OPSTATE_STORE_POLICY(pol) (pol)      // A/B: -DDS_OPSTATE_STORE_POLICY\(pol\)=0 keeps the stores ordinary while the loads stream
#endif
template <int POL> struct StRefT {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff, soff;
    __device__ operator float() const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, POL)); }
    __device__ void operator=(float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, DS_OPSTATE_STORE_POLICY(POL)); }
    __device__ void operator=(const StRefT& o) const { *this = (float)o; }       // st_at(..) = st_at(..) moves the value, not the reference
    StRefT(const StRefT&) = default;
    __device__ StRefT(__amdgpu_buffer_rsrc_t rs_, unsigned v, unsigned s) : rs(rs_), voff(v), soff(s) {}
};
typedef StRefT<DS_OPSTATE_POLICY> StRef;
__device__ inline OpCtx make_op_ctx(const OpParams& p0, long long first_lane) {       // first_lane: flat (b, k) index of the workgroup's lane 0
    OpCtx c;
    static_cast<OpParams&>(c) = p0;
    if (p0.dev_cnt) { c.frm_cnt = p0.dev_cnt[0]; c.ell = p0.dev_cnt[1]; c.first_frame = p0.dev_cnt[2]; }
    c.b0 = (int)(first_lane / p0.KP);
    const long long off = (long long)c.b0 * st_floats_per_bin(p0.NF) * p0.KP;
    const long long left = ((long long)p0.B * st_floats_per_bin(p0.NF) * p0.KP - off) * 4;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(p0.st + off, 0, (int)(unsigned)(left > 0xffffffffLL ? 0xffffffffLL : (left > 0 ? left : 0)), 0x00020000);
    return c;
}
__device__ inline StRef st_at(const OpCtx& p, int b, int f, int k) {
    // [b][f / 4][k][f % 4]: the lane offset addresses the bin's 16-byte group, the plane group travels in the SGPR offset and f % 4 is an
    // immediate — four neighbouring floats of a bin merge into one 16-byte access per lane
#ifdef DS_OLD_OPSTATE
    return StRef(p.rs, (unsigned)(((b - p.b0) * st_floats_per_bin(p.NF) * p.KP + k) * 4), (unsigned)(f * p.KP * 4));
#else
    return StRef(p.rs, (unsigned)((((b - p.b0) * st_floats_per_bin(p.NF) * p.KP + k * 4) + (f & 3)) * 4), (unsigned)((f >> 2) * p.KP * 16));
#endif
}
// the same word with the ordinary cache policy: for the operators that keep their state in memory and come back to it every frame of a
// call (OMLSA, the run-time-shape subband LMS) — a streamed line would be fetched from HBM again one frame later
__device__ inline StRefT<0> st_mem(const OpCtx& p, int b, int f, int k) {
    // [b][f / 4][k][f % 4]: the lane offset addresses the bin's 16-byte group, the plane group travels in the SGPR offset and f % 4 is an
    // immediate — four neighbouring floats of a bin merge into one 16-byte access per lane
#ifdef DS_OLD_OPSTATE
    return StRefT<0>(p.rs, (unsigned)(((b - p.b0) * st_floats_per_bin(p.NF) * p.KP + k) * 4), (unsigned)(f * p.KP * 4));
#else
    return StRefT<0>(p.rs, (unsigned)((((b - p.b0) * st_floats_per_bin(p.NF) * p.KP + k * 4) + (f & 3)) * 4), (unsigned)((f >> 2) * p.KP * 16));
#endif
}
// one whole float4 plane group q (floats 4 q .. 4 q + 3) of bin k as ONE 16-byte access
// (four dword loads at consecutive immediate offsets: the backend merges them into one buffer_load_dwordx4.  ROCm 7.2's
// __builtin_amdgcn_raw_buffer_load_b128 is lowered to a ONE-dword load — three result words undefined — so it is not used.  Neither is
// the b128 STORE builtin: see st_store4)
__device__ inline void st_load4(const OpCtx& p, int b, int q, int k, float* d) {
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = st_at(p, b, 4 * q + j, k);
}
// (four dword stores at consecutive immediate offsets, which the backend merges (dwordx3 + dword in ROCm 7.2).  The b128 STORE builtin
// was tried as well: single-stream results were right, but with two utterance groups of a chain running on two streams the outputs
// carried NaNs at random — scratch/dbg_groups.py, DESIGN.md section 3.5 — so neither b128 builtin is used)
// Round 4: ONE buffer_store_dwordx4 per group, written as inline assembly.  Left to the backend the four dword stores of a group came out
// as a dword + a dwordx3 (the float-0 store's offset register is formed from another spelling of the same stride; three rewrites of the
// expression each merged less): two partial writes per 16 bytes, which the L2 merged while the stores were ordinary and nothing merges now
// that they are streamed (+30 % write bytes in the PMC passes, profiles/r04i).  The trailing s_nop covers the hardware's two wait states
// between a store of more than 8 bytes and a vector write to its data registers, the leading one the five between a vector write to an
// SGPR (v_readfirstlane) and a memory instruction that reads it: hazards the compiler cannot see into an asm statement for.
// -DDS_OPSTATE_STORE4_BUILTIN: the four dword stores again (A/B)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(DS_OPSTATE_STORE4_BUILTIN) && !defined(DS_OLD_OPSTATE)
#error "st_store4 is hand-written gfx950 ISA with hand-counted wait states: this library builds for gfx950 only (make ARCH=gfx950)"
#endif
__device__ inline void st_store4(const OpCtx& p, int b, int q, int k, const float* s) {
#if defined(DS_OPSTATE_STORE4_BUILTIN) || defined(DS_OLD_OPSTATE)
#pragma unroll
    for (int j = 0; j < 4; ++j) st_at(p, b, 4 * q + j, k) = s[j];
#else
    typedef float v4_t __attribute__((ext_vector_type(4)));
    v4_t v; v.x = s[0]; v.y = s[1]; v.z = s[2]; v.w = s[3];
    const unsigned voff = (unsigned)(((b - p.b0) * st_floats_per_bin(p.NF) * p.KP + k * 4) * 4);
    const unsigned soff = (unsigned)(q * p.KP * 16);
#if DS_OPSTATE_POLICY == 2
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(p.rs), "s"(soff) : "memory");
#else
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" : : "v"(v), "v"(voff), "s"(p.rs), "s"(soff) : "memory");
#endif
#endif
}
#else
typedef OpParams OpCtx;
DS_HD OpCtx make_op_ctx(const OpParams& p0, long long) { return p0; }
DS_HD float& st_at(const OpCtx& p, int b, int f, int k) { return p.st[st_index(b, f, k, p.NF, p.KP)]; }
DS_HD float& st_mem(const OpCtx& p, int b, int f, int k) { return st_at(p, b, f, k); }
DS_HD void st_load4(const OpCtx& p, int b, int q, int k, float* d) { for (int j = 0; j < 4; ++j) d[j] = st_at(p, b, 4 * q + j, k); }
DS_HD void st_store4(const OpCtx& p, int b, int q, int k, const float* s) { for (int j = 0; j < 4; ++j) st_at(p, b, 4 * q + j, k) = s[j]; }
#endif
// floats F0 .. F0 + N - 1 of a bin into / out of registers: whole float4 groups as 16-byte accesses, the ragged ends float by float
// (compile-time recursion: every register index is a constant)
template <int F, int END, int F0> DS_HD void st_load_from(const OpCtx& p, int b, int k, float* d) {
    if constexpr (F < END) {
        if constexpr ((F & 3) == 0 && F + 4 <= END) { st_load4(p, b, F >> 2, k, d + (F - F0)); st_load_from<F + 4, END, F0>(p, b, k, d); }
        else { d[F - F0] = st_at(p, b, F, k); st_load_from<F + 1, END, F0>(p, b, k, d); }
    }
}
template <int F, int END, int F0> DS_HD void st_store_from(const OpCtx& p, int b, int k, const float* s) {
    if constexpr (F < END) {
        if constexpr ((F & 3) == 0 && F + 4 <= END) { st_store4(p, b, F >> 2, k, s + (F - F0)); st_store_from<F + 4, END, F0>(p, b, k, s); }
        else { st_at(p, b, F, k) = s[F - F0]; st_store_from<F + 1, END, F0>(p, b, k, s); }
    }
}
template <int F0, int N> DS_HD void st_load_span(const OpCtx& p, int b, int k, float* d) { st_load_from<F0, F0 + N, F0>(p, b, k, d); }
template <int F0, int N> DS_HD void st_store_span(const OpCtx& p, int b, int k, const float* s) { st_store_from<F0, F0 + N, F0>(p, b, k, s); }

// advance the uniform MCRA counters by one frame (mcra.py:52-56,72-74); returns `reset` for this frame
DS_HD bool mcra_tick(int& frm, int& ell, int L) {
    const bool reset = (frm != 0) && (ell % L == 0);
    if (reset) ell = 0;
    return reset;
}

// ------------------------------------------------------------------------------------------------
// MCRA: in0 = Y [B][T][K] power (or complex [B][T][K][2]), out0 = lambda_d [B][T][K], optional out1 = p [B][T][K].  state: S,Smin,Stmp,p,lambda_d
// ------------------------------------------------------------------------------------------------
DS_HD float mcra_in(const OpParams& p, long long base, int k) {
    if (p.in_complex) { const float re = p.in0[2 * (base + k)], im = p.in0[2 * (base + k) + 1]; return fma_(re, re, im * im); }
    return p.in0[base + k];
}

DS_HD void op_mcra(const OpCtx& p, int b, int k) {
    float st[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) st[f] = st_at(p, b, f, k);
    int frm = p.frm_cnt, ell = p.ell;
    for (int t = 0; t < p.T; ++t) {
        const long long base = ((long long)b * p.T + t) * p.K;
        const bool reset = mcra_tick(frm, ell, p.L);
        const float y0 = mcra_in(p, base, k);
        const float ym = k > 0 ? mcra_in(p, base, k - 1) : 0.0f;
        const float yp = k < p.K - 1 ? mcra_in(p, base, k + 1) : 0.0f;
        mcra_bin(st, k, p.K, ym, y0, yp, frm, reset, p.L);
        frm += 1; ell += 1;
        p.out0[base + k] = st[4];
        if (p.out1) p.out1[base + k] = st[3];                 // speech presence probability of this frame (mcra.p)
    }
#pragma unroll
    for (int f = 0; f < 5; ++f) st_at(p, b, f, k) = st[f];
}

// ------------------------------------------------------------------------------------------------
// McMcra: in0 = y complex [B][T][K][M]; out0 = p, out1 = G [B][T][K].
// state: Phi_yy, Phi_vv (packed symmetric), then xi, gamma, p, G of the last frame (attributes users read)
// ------------------------------------------------------------------------------------------------
template <int M> DS_HD void op_mcmcra(const OpCtx& p, int b, int k) {
    constexpr int NS = M * (M + 1) / 2;
    float both[2 * NS + 4];                                  // Phi_yy, Phi_vv (packed symmetric), then xi, gamma, p, G: the bin's whole state
    float* pyy = both;
    float* pvv = both + NS;
    st_load_span<0, 2 * NS>(p, b, k, both);
    float pp = 0, G = 0, xi = 0, gam = 0;
    int frm = p.frm_cnt;
    for (int t = 0; t < p.T; ++t) {
        const long long base = (((long long)b * p.T + t) * p.K + k) * M;
        cf Z[M];
#pragma unroll
        for (int m = 0; m < M; ++m) Z[m] = mk(p.in0[2 * (base + m)], p.in0[2 * (base + m) + 1]);
        mcmcra_bin<M>(pyy, pvv, Z, k, frm, pp, G, xi, gam);
        frm += 1;
        const long long ob = ((long long)b * p.T + t) * p.K + k;
        p.out0[ob] = pp;
        p.out1[ob] = G;
    }
    if (p.T > 0) {
        both[2 * NS + 0] = xi; both[2 * NS + 1] = gam; both[2 * NS + 2] = pp; both[2 * NS + 3] = G;
        st_store_span<0, 2 * NS + 4>(p, b, k, both);
    } else {
        st_store_span<0, 2 * NS>(p, b, k, both);
    }
}

// ------------------------------------------------------------------------------------------------
// NsOmlsaMulti: in0 = y [B][T][K] (beam power), in1 = u [B][T][K][M-1] (reference powers) — or, with in_complex, the complex
// spectra themselves (|.|^2 formed here, TDGSC.py:162-163); out0 = lambda_d, out1 = G, out2 = p  [B][T][K]; with in_complex
// optional out3 = Y * sqrt(G) complex [B][T][K], the post-filtered beam spectrum (TDGSC.py:166-168).
// state floats: [0,5M) M MCRAs (beam first), then zeta_Y, zeta_U[M-1], lambda_d, gamma, G_H1, G, p, xi_hat, q_hat
// ------------------------------------------------------------------------------------------------
DS_HD int omlsa_nf(int M) { return 5 * M + (M - 1) + 8; }

DS_HD void op_omlsa(const OpCtx& p, int b, int k) {
    const int M = p.M, K = p.K, R = M - 1;
    const int o_zy = 5 * M, o_zu = 5 * M + 1, o_s = 5 * M + 1 + R;   // o_s: lambda_d, gamma, G_H1, G, p, xi_hat, q_hat
    int frm = p.frm_cnt, ell = p.ell, first = p.first_frame;
    const float Gmin = 0.06309573444801933f;               // 10^(-12/10)  (omlsa_multi.py:35-36)
    for (int t = 0; t < p.T; ++t) {
        const long long yb = ((long long)b * p.T + t) * K;
        const long long ub = yb * R;
        const bool reset = mcra_tick(frm, ell, p.L);
        auto pw = [&](const float* src, long long i) {          // power of element i of a real-power or complex-spectrum array
            return p.in_complex ? fma_(src[2 * i], src[2 * i], src[2 * i + 1] * src[2 * i + 1]) : src[i];
        };
        const float y0 = pw(p.in0, yb + k);
        const float ym = k > 0 ? pw(p.in0, yb + k - 1) : 0.0f, yp = k < K - 1 ? pw(p.in0, yb + k + 1) : 0.0f;
        // M minima-controlled noise trackers (:83-85)
        float mc[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) mc[f] = st_mem(p, b, f, k);
        mcra_bin(mc, k, K, ym, y0, yp, frm, reset, p.L);
#pragma unroll
        for (int f = 0; f < 5; ++f) st_mem(p, b, f, k) = mc[f];
        const float MU_Y = mc[4];
        float zu_minus_mu_max = -3.0e38f;
        for (int ch = 0; ch < R; ++ch) {
            const float u0 = pw(p.in1, ub + (long long)k * R + ch);
            const float um = k > 0 ? pw(p.in1, ub + (long long)(k - 1) * R + ch) : 0.0f;
            const float up = k < K - 1 ? pw(p.in1, ub + (long long)(k + 1) * R + ch) : 0.0f;
#pragma unroll
            for (int f = 0; f < 5; ++f) mc[f] = st_mem(p, b, 5 * (ch + 1) + f, k);
            mcra_bin(mc, k, K, um, u0, up, frm, reset, p.L);
#pragma unroll
            for (int f = 0; f < 5; ++f) st_mem(p, b, 5 * (ch + 1) + f, k) = mc[f];
            float zu;
            if (first) zu = u0;                                                           // :92-93
            else zu = fma_(0.8f, st_mem(p, b, o_zu + ch, k), (float)(1.0 - 0.8) * fma_(up, 0.25f, fma_(u0, 0.5f, um * 0.25f)));   // :100
            st_mem(p, b, o_zu + ch, k) = zu;
            zu_minus_mu_max = fmaxf_(zu_minus_mu_max, zu - mc[4]);
        }
        frm += 1; ell += 1;
        const long long ob = yb + k;
        if (first) {                                                                       // :87-93
            first = 0;
            st_mem(p, b, o_s + 0, k) = y0;
            st_mem(p, b, o_zy, k) = y0;
            p.out0[ob] = y0; p.out1[ob] = st_mem(p, b, o_s + 3, k); p.out2[ob] = st_mem(p, b, o_s + 4, k);
            if (p.in_complex && p.out3) {
                const float sg = sqrtf(st_mem(p, b, o_s + 3, k));
                p.out3[2 * ob] = p.in0[2 * ob] * sg; p.out3[2 * ob + 1] = p.in0[2 * ob + 1] * sg;
            }
            continue;
        }
        const float zy = fma_(0.8f, st_mem(p, b, o_zy, k), (float)(1.0 - 0.8) * fma_(yp, 0.25f, fma_(y0, 0.5f, ym * 0.25f)));   // :98
        st_mem(p, b, o_zy, k) = zy;
        float Omega = fmaxf_(zy - MU_Y, 1e-6f) / (fmaxf_(zu_minus_mu_max, 0.01f * MU_Y) + 1e-6f);   // :107-109
        Omega = fminf_(fmaxf_(Omega, 0.1f), 100.0f);
        const float gamma_s = fminf_(y0 / fma_(MU_Y, 1.66f, 1e-6f), 100.0f);                     // :115
        // :122-130.  The reference clips q to [1e-6, 0.9999998] and then uses q / (1 - q); in fp32 1 - 0.9999998f is off by
        // 10 %, so the complement 1 - q = min((gamma_s - 1) / 9, (Omega - 0.3) / 2.7) is formed directly and clipped instead.
        float omq;
        if (gamma_s < 1.0f || Omega < 0.3f) omq = 0.0f;
        else omq = fminf_((gamma_s - 1.0f) / (10.0f - 1.0f), (Omega - 0.3f) / (3.0f - 0.3f));
        omq = fminf_(fmaxf_(omq, 2e-7f), 1.0f - 1e-6f);
        const float q = 1.0f - omq;
        float lam = st_mem(p, b, o_s + 0, k);
        const float gamma_pre = st_mem(p, b, o_s + 1, k), gh1_pre = st_mem(p, b, o_s + 2, k);
        const float gamma = y0 / fmaxf_(lam, 1e-10f);                                            // :134
        const float xi = fma_(0.921f * gh1_pre * gh1_pre, gamma_pre, (float)(1.0 - 0.921) * fmaxf_(gamma - 1.0f, 0.0f));   // :137
        const float nu = gamma * xi / (1.0f + xi);                                               // :140
        const float gh1 = xi / (1.0f + xi);                                                      // :144
        const float pp = 1.0f / (1.0f + (1.0f - omq) / omq * (1.0f + xi) * expf(-nu));           // :147
        const float at = fma_((float)(1.0 - 0.85), pp, 0.85f);                                   // Base :57, alpha_d = 0.85
        lam = fma_(at, lam, 1.47f * (1.0f - at) * y0);                                           // :149
        float G = powf(gh1, pp) * powf(Gmin, 1.0f - pp);                                         // :153
        G = fmaxf_(fminf_(G, 1.0f), Gmin);
        st_mem(p, b, o_s + 0, k) = lam; st_mem(p, b, o_s + 1, k) = gamma; st_mem(p, b, o_s + 2, k) = gh1;
        st_mem(p, b, o_s + 3, k) = G; st_mem(p, b, o_s + 4, k) = pp; st_mem(p, b, o_s + 5, k) = xi; st_mem(p, b, o_s + 6, k) = q;
        p.out0[ob] = lam; p.out1[ob] = G; p.out2[ob] = pp;
        if (p.in_complex && p.out3) {
            const float sg = sqrtf(G);
            p.out3[2 * ob] = p.in0[2 * ob] * sg; p.out3[2 * ob + 1] = p.in0[2 * ob + 1] * sg;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Subband (N)LMS, single- or multi-channel: in0 = x complex [B][T][K][C], in1 = d complex [B][T][K],
// in2 = p [B][T][K] (optional); out0 = err complex [B][T][K].
// state floats: W [N][C] complex, input_buffer [N][C] complex, P
// ------------------------------------------------------------------------------------------------
DS_HD int sublms_nf(int N, int C) { return 4 * N * C + 1; }

// desired-signal sample of instance b, frame t, bin k (see OpParams::d_interleaved / d_prev)
DS_HD cf subband_d(const OpParams& p, int b, int t, int k) {
    if (p.d_prev) {
        if (t == 0) return mk(p.d_prev[2 * ((long long)b * p.K + k)], p.d_prev[2 * ((long long)b * p.K + k) + 1]);
        t -= 1;
    }
    long long i;
    if (p.d_interleaved) i = (((long long)(b / p.x_fan) * p.T + t) * p.K + k) * p.x_fan + b % p.x_fan;
    else i = ((long long)b * p.T + t) * p.K + k;
    return mk(p.in1[2 * i], p.in1[2 * i + 1]);
}
DS_HD void subband_d_carry(const OpCtx& p, int b, int k) {        // after the T frames: keep the last frame of in1 for the next call
    if (p.d_prev && p.T > 0) {
        const long long i = ((long long)b * p.T + p.T - 1) * p.K + k;
        p.d_prev[2 * ((long long)b * p.K + k)] = p.in1[2 * i];
        p.d_prev[2 * ((long long)b * p.K + k) + 1] = p.in1[2 * i + 1];
    }
}

DS_HD void op_sublms_generic(const OpCtx& p, int b, int k) {
    const int N = p.N, C = p.M, NC2 = 2 * N * C;
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K + k;
        const long long fx = ((long long)(b / p.x_fan) * p.T + t) * p.K + k;
        // shift register (SubbandAF.py:50-51 / SubbandLmsMc.py:62-63)
        for (int n = N - 1; n > 0; --n)
            for (int c = 0; c < 2 * C; ++c) st_mem(p, b, NC2 + n * 2 * C + c, k) = st_mem(p, b, NC2 + (n - 1) * 2 * C + c, k);
        for (int c = 0; c < C; ++c) {
            st_mem(p, b, NC2 + 2 * c, k) = p.in0[2 * (fx * C + c)];
            st_mem(p, b, NC2 + 2 * c + 1, k) = p.in0[2 * (fx * C + c) + 1];
        }
        cf out = mk(0.0f, 0.0f);
        float pw = 0.0f;
        for (int i = 0; i < N * C; ++i) {
            const cf w = mk(st_mem(p, b, 2 * i, k), st_mem(p, b, 2 * i + 1, k));
            const cf x = mk(st_mem(p, b, NC2 + 2 * i, k), st_mem(p, b, NC2 + 2 * i + 1, k));
            out = cfmac(out, x, w);                        // conj(W) X  (SubbandAF.py:107)
            pw += cabs2(x);
        }
        float pk = p.has_p ? p.in2[fx] : 1.0f;
        if (p.p_complement) pk = 1.0f - pk;
        const cf d = subband_d(p, b, t, k);
        const cf err = mk(fma_(-out.x, pk, d.x), fma_(-out.y, pk, d.y));      // d - out * p  (SubbandLMS.py:66-68)
        float scale = 1.0f;
        if (p.norm) {
            float P = st_mem(p, b, 2 * NC2, k);
            P = fma_(p.alpha, P, (1.0f - p.alpha) * (pw / (float)C));           // SubbandLMS.py:72-75 ; /M in SubbandLmsMc.py:174-180
            st_mem(p, b, 2 * NC2, k) = P;
            scale = 1.0f / (P + p.reg);
        }
        const float g = 2.0f * p.mu * pk * scale;                              // SubbandAF.py:86
        for (int i = 0; i < N * C; ++i) {
            const cf x = mk(st_mem(p, b, NC2 + 2 * i, k), st_mem(p, b, NC2 + 2 * i + 1, k));
            const cf gr = cmulc(x, err);                                        // X conj(err)
            st_mem(p, b, 2 * i, k) = fma_(g, gr.x, st_mem(p, b, 2 * i, k));
            st_mem(p, b, 2 * i + 1, k) = fma_(g, gr.y, st_mem(p, b, 2 * i + 1, k));
        }
        p.out0[2 * fb] = err.x; p.out0[2 * fb + 1] = err.y;
    }
    subband_d_carry(p, b, k);
}

// the same recursion with W, the tap buffer and P held in registers for the T frames of the call (state read once, written once)
template <int N, int C> DS_HD void op_sublms_t(const OpCtx& p, int b, int k) {
    constexpr int NC = N * C, NC2 = 2 * NC;
    cf W[NC], X[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        W[i] = mk(st_at(p, b, 2 * i, k), st_at(p, b, 2 * i + 1, k));
        X[i] = mk(st_at(p, b, NC2 + 2 * i, k), st_at(p, b, NC2 + 2 * i + 1, k));
    }
    float P = st_at(p, b, 2 * NC2, k);
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K + k;
        const long long fx = ((long long)(b / p.x_fan) * p.T + t) * p.K + k;
#pragma unroll
        for (int n = N - 1; n > 0; --n)
#pragma unroll
            for (int c = 0; c < C; ++c) X[n * C + c] = X[(n - 1) * C + c];
#pragma unroll
        for (int c = 0; c < C; ++c) X[c] = mk(p.in0[2 * (fx * C + c)], p.in0[2 * (fx * C + c) + 1]);
        cf out = mk(0.0f, 0.0f);
        float pw = 0.0f;
#pragma unroll
        for (int i = 0; i < NC; ++i) { out = cfmac(out, X[i], W[i]); pw += cabs2(X[i]); }
        float pk = p.has_p ? p.in2[fx] : 1.0f;
        if (p.p_complement) pk = 1.0f - pk;
        const cf d = subband_d(p, b, t, k);
        const cf err = mk(fma_(-out.x, pk, d.x), fma_(-out.y, pk, d.y));
        float scale = 1.0f;
        if (p.norm) {
            P = fma_(p.alpha, P, (1.0f - p.alpha) * (pw / (float)C));
            scale = 1.0f / (P + p.reg);
        }
        const float g = 2.0f * p.mu * pk * scale;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const cf gr = cmulc(X[i], err);
            W[i] = mk(fma_(g, gr.x, W[i].x), fma_(g, gr.y, W[i].y));
        }
        p.out0[2 * fb] = err.x; p.out0[2 * fb + 1] = err.y;
    }
    subband_d_carry(p, b, k);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        st_at(p, b, 2 * i, k) = W[i].x; st_at(p, b, 2 * i + 1, k) = W[i].y;
        st_at(p, b, NC2 + 2 * i, k) = X[i].x; st_at(p, b, NC2 + 2 * i + 1, k) = X[i].y;
    }
    if (p.norm) st_at(p, b, 2 * NC2, k) = P;
}

// The fan form of the single-channel subband LMS (see op_subrls_fan further down for the idea): the F instances of an utterance share
// the reference input, hence the tap buffer, the smoothed input power and the step size; only the error and the weights are per instance.
// Same arithmetic per instance as op_sublms_t<N, 1>.  X and P live in the first instance's planes.  Needs d_interleaved, no d_prev.
template <int N, int F> DS_HD void op_sublms_fan(const OpCtx& p, int u, int k) {
    constexpr int NC2 = 2 * N;
    const int b0 = u * F;
    cf W[F][N], X[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        X[i] = mk(st_at(p, b0, NC2 + 2 * i, k), st_at(p, b0, NC2 + 2 * i + 1, k));
#pragma unroll
        for (int m = 0; m < F; ++m) W[m][i] = mk(st_at(p, b0 + m, 2 * i, k), st_at(p, b0 + m, 2 * i + 1, k));
    }
    float P = st_at(p, b0, 2 * NC2, k);
    for (int t = 0; t < p.T; ++t) {
        const long long fx = ((long long)u * p.T + t) * p.K + k;
#pragma unroll
        for (int n = N - 1; n > 0; --n) X[n] = X[n - 1];
        X[0] = mk(p.in0[2 * fx], p.in0[2 * fx + 1]);
        cf d[F];
#pragma unroll
        for (int m = 0; m < F; ++m) d[m] = mk(p.in1[2 * (fx * F + m)], p.in1[2 * (fx * F + m) + 1]);
        float pw = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) pw += cabs2(X[i]);
        float pk = p.has_p ? p.in2[fx] : 1.0f;
        if (p.p_complement) pk = 1.0f - pk;
        float scale = 1.0f;
        if (p.norm) {
            P = fma_(p.alpha, P, (1.0f - p.alpha) * pw);                        // one channel: pw / C = pw
            scale = 1.0f / (P + p.reg);
        }
        const float g = 2.0f * p.mu * pk * scale;
#pragma unroll
        for (int m = 0; m < F; ++m) {
            cf out = mk(0.0f, 0.0f);
#pragma unroll
            for (int i = 0; i < N; ++i) out = cfmac(out, X[i], W[m][i]);
            const cf err = mk(fma_(-out.x, pk, d[m].x), fma_(-out.y, pk, d[m].y));
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const cf gr = cmulc(X[i], err);
                W[m][i] = mk(fma_(g, gr.x, W[m][i].x), fma_(g, gr.y, W[m][i].y));
            }
            const long long fb = ((long long)(b0 + m) * p.T + t) * p.K + k;
            p.out0[2 * fb] = err.x; p.out0[2 * fb + 1] = err.y;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int m = 0; m < F; ++m) { st_at(p, b0 + m, 2 * i, k) = W[m][i].x; st_at(p, b0 + m, 2 * i + 1, k) = W[m][i].y; }
        st_at(p, b0, NC2 + 2 * i, k) = X[i].x; st_at(p, b0, NC2 + 2 * i + 1, k) = X[i].y;
    }
    if (p.norm) st_at(p, b0, 2 * NC2, k) = P;
}
// can a SubbandLMS call run as op_sublms_fan?
inline bool sublms_fan_ok(const OpParams& p) {
    return p.N == 2 && p.M == 1 && p.d_interleaved && p.d_prev == nullptr && (p.x_fan == 2 || p.x_fan == 4 || p.x_fan == 6 || p.x_fan == 8) && p.B % p.x_fan == 0;
}

DS_HD void op_sublms(const OpCtx& p, int b, int k) {
    if (p.N == 2) {                                   // the reference's tap count everywhere it builds these filters
        switch (p.M) {
            case 1: return op_sublms_t<2, 1>(p, b, k);
            case 2: return op_sublms_t<2, 2>(p, b, k);
            case 4: return op_sublms_t<2, 4>(p, b, k);
            case 6: return op_sublms_t<2, 6>(p, b, k);
            case 8: return op_sublms_t<2, 8>(p, b, k);
        }
    }
    op_sublms_generic(p, b, k);
}

// ------------------------------------------------------------------------------------------------
// Subband RLS: in0 = x complex [B][T][K], in1 = d complex [B][T][K]; out0 = err complex [B][T][K].
// state floats: W [N] complex, input_buffer [N] complex, P [N][N] complex
// ------------------------------------------------------------------------------------------------
constexpr int RLS_NMAX = 4;
DS_HD int subrls_nf(int N) { return 4 * N + 2 * N * N; }

template <int N> DS_HD void op_subrls_t(const OpCtx& p, int b, int k) {
    constexpr int oX = 2 * N, oP = 4 * N;
    const float lam_inv = 1.0f / p.lam;
    cf W[N], X[N], P[N][N];                              // in registers for the T frames of the call
#pragma unroll
    for (int i = 0; i < N; ++i) {
        W[i] = mk(st_at(p, b, 2 * i, k), st_at(p, b, 2 * i + 1, k));
        X[i] = mk(st_at(p, b, oX + 2 * i, k), st_at(p, b, oX + 2 * i + 1, k));
#pragma unroll
        for (int j = 0; j < N; ++j) P[i][j] = mk(st_at(p, b, oP + 2 * (i * N + j), k), st_at(p, b, oP + 2 * (i * N + j) + 1, k));
    }
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K + k;
        const long long fx = ((long long)(b / p.x_fan) * p.T + t) * p.K + k;
#pragma unroll
        for (int n = N - 1; n > 0; --n) X[n] = X[n - 1];
        X[0] = mk(p.in0[2 * fx], p.in0[2 * fx + 1]);
        cf num[N], xhP[N];
        cf out = mk(0.0f, 0.0f);
#pragma unroll
        for (int i = 0; i < N; ++i) out = cfmac(out, X[i], W[i]);
        const cf d = subband_d(p, b, t, k);
        const cf err = csub(d, out);                                             // SubbandRLS.py:52
        cf den = mk(p.lam, 0.0f);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            cf a = mk(0.0f, 0.0f), r = mk(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                a = cfma(a, P[i][j], X[j]);                                       // (P x)_i
                r = cfmac(r, P[j][i], X[j]);                                      // (x^H P)_i
            }
            num[i] = a; xhP[i] = r;
            den = cfmac(den, a, X[i]);                                            // + conj(x_i) (P x)_i   :56-60
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const cf kn = cdiv(num[i], den);
#pragma unroll
            for (int j = 0; j < N; ++j) P[i][j] = cscale(cfnma(P[i][j], kn, xhP[j]), lam_inv);   // P = (P - kn x^H P) / lambda   :63
            const cf g = cmulc(kn, err);                                         // conj(err) kn   :65
            W[i] = mk(fma_(2.0f * p.mu, g.x, W[i].x), fma_(2.0f * p.mu, g.y, W[i].y));
        }
        p.out0[2 * fb] = err.x; p.out0[2 * fb + 1] = err.y;
    }
    subband_d_carry(p, b, k);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        st_at(p, b, 2 * i, k) = W[i].x; st_at(p, b, 2 * i + 1, k) = W[i].y;
        st_at(p, b, oX + 2 * i, k) = X[i].x; st_at(p, b, oX + 2 * i + 1, k) = X[i].y;
#pragma unroll
        for (int j = 0; j < N; ++j) { st_at(p, b, oP + 2 * (i * N + j), k) = P[i][j].x; st_at(p, b, oP + 2 * (i * N + j) + 1, k) = P[i][j].y; }
    }
}

// The F = x_fan instances of an utterance (SubbandGSC's blocking filters) take the same reference input, so their tap buffers, their
// inverse correlation matrices P and their gain vectors are the same numbers: one thread per (utterance, bin) runs that common part once
// and the F error / weight updates after it — the arithmetic of every instance is exactly op_subrls_t's, in the same order.  X and P
// live in the first instance's planes only: the X / P planes of the other F - 1 instances are not maintained (this kernel is their only
// reader; the chain exports and imports the raw state, so checkpoints round-trip).
// Needs d_interleaved (the desired signals are the channels of one spectrum) and no d_prev.
// the fan's state and one frame of it (shared by op_subrls_fan and the fused OP_MCSPP_STEADY_FAN: the same statements, the same results)
template <int N, int F> struct RlsFan {
    static constexpr int oX = 2 * N, oP = 4 * N;
    cf W[F][N], X[N], P[N][N];
    // rows 2 i, 2 i + 1 of an instance are (Re, Im) of tap i: a cf array IS the run of rows it mirrors — whole float4 groups per access
    DS_HD void load(const OpCtx& p, int b0, int k) {
        st_load_span<oX, 2 * N>(p, b0, k, reinterpret_cast<float*>(&X[0]));
        st_load_span<oP, 2 * N * N>(p, b0, k, reinterpret_cast<float*>(&P[0][0]));
#pragma unroll
        for (int m = 0; m < F; ++m) st_load_span<0, 2 * N>(p, b0 + m, k, reinterpret_cast<float*>(&W[m][0]));
    }
    DS_HD void store(const OpCtx& p, int b0, int k) const {
#pragma unroll
        for (int m = 0; m < F; ++m) st_store_span<0, 2 * N>(p, b0 + m, k, reinterpret_cast<const float*>(&W[m][0]));
        st_store_span<oX, 2 * N>(p, b0, k, reinterpret_cast<const float*>(&X[0]));
        st_store_span<oP, 2 * N * N>(p, b0, k, reinterpret_cast<const float*>(&P[0][0]));
    }
    // the state to / from a per-lane parking area (LDS on the device; element f of this lane at area[f * stride]): the fused operator keeps
    // the filters there between their frames, so that they cost McSpp's estimation core no registers
    static constexpr int NW = 2 * N * (F + 1 + N);
    DS_HD void park(float* area, int stride) const {
        const float* w = reinterpret_cast<const float*>(&W[0][0]);
#pragma unroll
        for (int f = 0; f < 2 * N * F; ++f) area[f * stride] = w[f];
        const float* x = reinterpret_cast<const float*>(&X[0]);
#pragma unroll
        for (int f = 0; f < 2 * N; ++f) area[(2 * N * F + f) * stride] = x[f];
        const float* q = reinterpret_cast<const float*>(&P[0][0]);
#pragma unroll
        for (int f = 0; f < 2 * N * N; ++f) area[(2 * N * F + 2 * N + f) * stride] = q[f];
    }
    DS_HD void unpark(const float* area, int stride) {
        float* w = reinterpret_cast<float*>(&W[0][0]);
#pragma unroll
        for (int f = 0; f < 2 * N * F; ++f) w[f] = area[f * stride];
        float* x = reinterpret_cast<float*>(&X[0]);
#pragma unroll
        for (int f = 0; f < 2 * N; ++f) x[f] = area[(2 * N * F + f) * stride];
        float* q = reinterpret_cast<float*>(&P[0][0]);
#pragma unroll
        for (int f = 0; f < 2 * N * N; ++f) q[f] = area[(2 * N * F + 2 * N + f) * stride];
    }
    // one frame: reference sample x, the F desired samples d -> the F errors (SubbandRLS.py:44-71 for each of the F filters)
    DS_HD void step(cf x, const cf* d, float lam, float mu, cf* err_out) {
        const float lam_inv = 1.0f / lam;
#pragma unroll
        for (int n = N - 1; n > 0; --n) X[n] = X[n - 1];
        X[0] = x;
        cf num[N], xhP[N], kn[N];
        cf den = mk(lam, 0.0f);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            cf a = mk(0.0f, 0.0f), r = mk(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                a = cfma(a, P[i][j], X[j]);
                r = cfmac(r, P[j][i], X[j]);
            }
            num[i] = a; xhP[i] = r;
            den = cfmac(den, a, X[i]);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            kn[i] = cdiv(num[i], den);
#pragma unroll
            for (int j = 0; j < N; ++j) P[i][j] = cscale(cfnma(P[i][j], kn[i], xhP[j]), lam_inv);
        }
#pragma unroll
        for (int m = 0; m < F; ++m) {
            cf out = mk(0.0f, 0.0f);
#pragma unroll
            for (int i = 0; i < N; ++i) out = cfmac(out, X[i], W[m][i]);
            const cf err = csub(d[m], out);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const cf g = cmulc(kn[i], err);
                W[m][i] = mk(fma_(2.0f * mu, g.x, W[m][i].x), fma_(2.0f * mu, g.y, W[m][i].y));
            }
            err_out[m] = err;
        }
    }
};
template <int N, int F> DS_HD void op_subrls_fan(const OpCtx& p, int u, int k) {
    const int b0 = u * F;
    RlsFan<N, F> fan;
    fan.load(p, b0, k);
    for (int t = 0; t < p.T; ++t) {
        const long long fx = ((long long)u * p.T + t) * p.K + k;
        cf d[F], err[F];
#pragma unroll
        for (int m = 0; m < F; ++m) d[m] = mk(p.in1[2 * (fx * F + m)], p.in1[2 * (fx * F + m) + 1]);
        fan.step(mk(p.in0[2 * fx], p.in0[2 * fx + 1]), d, p.lam, p.mu, err);
#pragma unroll
        for (int m = 0; m < F; ++m) {
            const long long fb = ((long long)(b0 + m) * p.T + t) * p.K + k;
            p.out0[2 * fb] = err[m].x; p.out0[2 * fb + 1] = err[m].y;
        }
    }
    fan.store(p, b0, k);
}
// can a SubbandRLS call run as op_subrls_fan?
inline bool subrls_fan_ok(const OpParams& p) {
    return p.N == 2 && p.d_interleaved && p.d_prev == nullptr && p.B % (p.x_fan > 0 ? p.x_fan : 1) == 0 && (p.x_fan == 2 || p.x_fan == 4 || p.x_fan == 6 || p.x_fan == 8);
}

DS_HD void op_subrls(const OpCtx& p, int b, int k) {
    switch (p.N) {
        case 1: return op_subrls_t<1>(p, b, k);
        case 2: return op_subrls_t<2>(p, b, k);
        case 3: return op_subrls_t<3>(p, b, k);
        default: return op_subrls_t<4>(p, b, k);
    }
}

// ------------------------------------------------------------------------------------------------
// McSppBase.estimation + compute_pmwf_weight (noise_estimation/mcspp_base.py:262-324,220-240):
// in0 = y complex [B][T][K][M]; out0 = p [B][T][K], out1 = w complex [B][T][K][M] (PMWF weights).
// state floats: Phi_yy (Hermitian packed: M diag + M(M-1)/2 complex), Phi_vv (same), MCRA(5), xi, gamma, p, w[M] complex
// ------------------------------------------------------------------------------------------------
DS_HD int mcsppbase_nf(int M) { return 2 * M * M + 5 + 3 + 2 * M; }

template <int M> DS_HD void op_mcsppbase(const OpCtx& p, int b, int k) {
    constexpr int NS = M * (M + 1) / 2, NO = M * (M - 1) / 2;
    float yd[M], yo[2 * NO + 1], vd[M], vo[2 * NO + 1], mc[5];
#pragma unroll
    for (int f = 0; f < M; ++f) { yd[f] = st_at(p, b, f, k); vd[f] = st_at(p, b, M * M + f, k); }
#pragma unroll
    for (int f = 0; f < 2 * NO; ++f) { yo[f] = st_at(p, b, M + f, k); vo[f] = st_at(p, b, M * M + M + f, k); }
#pragma unroll
    for (int f = 0; f < 5; ++f) mc[f] = st_at(p, b, 2 * M * M + f, k);
    int frm = p.frm_cnt, ell = p.ell;
    float xi = 0, gam = 0, pp = 0;
    cf w[M];
#pragma unroll
    for (int m = 0; m < M; ++m) w[m] = mk(0.0f, 0.0f);
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K;
        const long long base = (fb + k) * M;
        cf Z[M];
#pragma unroll
        for (int m = 0; m < M; ++m) Z[m] = mk(p.in0[2 * (base + m)], p.in0[2 * (base + m) + 1]);
        herm_rank1<M>(yd, yo, Z, 0.92f, (float)(1.0 - 0.92));                 // estimate_noisy_psd :86-92
        // real-symmetric inverse of Re(Phi_vv) + 1e-6 I  (:277-279)
        float sv[NS], sx[NS];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) {
                const int q = sym_index(i, j, M);
                if (i == j) { sv[q] = vd[i]; sx[q] = yd[i] - vd[i]; }
                else { const int o = off_index(i, j, M); sv[q] = vo[2 * o]; sx[q] = yo[2 * o] - vo[2 * o]; }
            }
        float Lm[M][M], inv_d[M], Li[M][M], iv[NS];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            float s = sym_get<M>(sv, j, j) + 1e-6f;
#pragma unroll
            for (int q = 0; q < j; ++q) s = fma_(-Lm[j][q], Lm[j][q], s);
            s = pivot_max(s, pivot_floor(1e-6f));
            const float r = 1.0f / sqrtf(s);
            inv_d[j] = r; Lm[j][j] = s * r;
#pragma unroll
            for (int i = j + 1; i < M; ++i) {
                float tt = sym_get<M>(sv, i, j);
#pragma unroll
                for (int q = 0; q < j; ++q) tt = fma_(-Lm[i][q], Lm[j][q], tt);
                Lm[i][j] = tt * r;
            }
        }
#pragma unroll
        for (int c = 0; c < M; ++c)
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (i < c) { Li[i][c] = 0.0f; continue; }
                float tt = (i == c) ? 1.0f : 0.0f;
#pragma unroll
                for (int q = c; q < i; ++q) tt = fma_(-Lm[i][q], Li[q][c], tt);
                Li[i][c] = tt * inv_d[i];
            }
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) {
                float tt = 0.0f;
#pragma unroll
                for (int q = j; q < M; ++q) tt = fma_(Li[q][i], Li[q][j], tt);
                iv[sym_index(i, j, M)] = tt;
            }
        // xi = trace(inv Re(Phi_xx)) ; gamma = Re(y^H inv Re(Phi_xx) inv y)  (:281-285) ; v = inv y
        float tr = 0.0f;
        cf v[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cf a = mk(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const float e = sym_get<M>(iv, i, j);
                tr = fma_(e, sym_get<M>(sx, i, j), tr);
                a.x = fma_(e, Z[j].x, a.x); a.y = fma_(e, Z[j].y, a.y);
            }
            v[i] = a;
        }
        float g = 0.0f;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j)
                g = fma_(sym_get<M>(sx, i, j), fma_(v[i].x, v[j].x, v[i].y * v[j].y), g);   // Re(conj(v_i) v_j) Pxx_ij
        xi = fminf_(fmaxf_(tr, 1e-6f), 1e6f);                                  // :287
        gam = fminf_(fmaxf_(g, 1e-6f), 1e6f);                                  // :288
        // compute_q (:96-118): MCRA on |y_0|^2, q = sqrt(1 - p_mcra)
        const bool reset = mcra_tick(frm, ell, p.L);
        const float y0 = cabs2(Z[0]);
        float ym = 0.0f, yp = 0.0f;
        if (k > 0) { const long long q = (fb + k - 1) * M; ym = cabs2(mk(p.in0[2 * q], p.in0[2 * q + 1])); }
        if (k < p.K - 1) { const long long q = (fb + k + 1) * M; yp = cabs2(mk(p.in0[2 * q], p.in0[2 * q + 1])); }
        mcra_bin(mc, k, p.K, ym, y0, yp, frm, reset, p.L);
        frm += 1; ell += 1;
        float q = sqrtf(1.0f - mc[3]);
        q = fminf_(fmaxf_(q, 0.01f), 0.99f);
        pp = 1.0f / (1.0f + q / (1.0f - q) * (1.0f + xi) * expf(-1.0f * (gam / (1.0f + xi))));   // :120-135
        pp = fminf_(fmaxf_(pp, 0.01f), 0.99f);
        // PMWF weights from Phi_xx (complex, before the noise update) and the real inverse (:220-240, :293)
        const float wsc = 1.0f / (1.0f + xi);
#pragma unroll
        for (int i = 0; i < M; ++i) {
            cf a = mk(0.0f, 0.0f);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const cf x0 = csub(herm_get<M>(yd, yo, j, 0), herm_get<M>(vd, vo, j, 0));   // Phi_xx[j][0]
                const float e = sym_get<M>(iv, i, j);
                a.x = fma_(e, x0.x, a.x); a.y = fma_(e, x0.y, a.y);
            }
            w[i] = cscale(a, wsc);
        }
        // update_noise_psd (:299-324): alpha_d = 0.92
        const float at = fma_((float)(1.0 - 0.92), pp, 0.92f);
        herm_rank1<M>(vd, vo, Z, at, 1.0f - at);
        const long long ob = fb + k;
        p.out0[ob] = pp;
#pragma unroll
        for (int m = 0; m < M; ++m) { p.out1[2 * (ob * M + m)] = w[m].x; p.out1[2 * (ob * M + m) + 1] = w[m].y; }
    }
#pragma unroll
    for (int f = 0; f < M; ++f) { st_at(p, b, f, k) = yd[f]; st_at(p, b, M * M + f, k) = vd[f]; }
#pragma unroll
    for (int f = 0; f < 2 * NO; ++f) { st_at(p, b, M + f, k) = yo[f]; st_at(p, b, M * M + M + f, k) = vo[f]; }
#pragma unroll
    for (int f = 0; f < 5; ++f) st_at(p, b, 2 * M * M + f, k) = mc[f];
    if (p.T > 0) {
        const int o = 2 * M * M + 5;
        st_at(p, b, o, k) = xi; st_at(p, b, o + 1, k) = gam; st_at(p, b, o + 2, k) = pp;
#pragma unroll
        for (int m = 0; m < M; ++m) { st_at(p, b, o + 3 + 2 * m, k) = w[m].x; st_at(p, b, o + 4 + 2 * m, k) = w[m].y; }
    }
}

// Small complex-Hermitian linear algebra in registers (inverse, principal eigenvector, MVDR weights): ds_linalg64.hpp (double)
template <int M> DS_HD void herm_unpack(const float* d, const float* o, cf (&A)[M][M]) {
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) A[i][j] = herm_get<M>(d, o, i, j);
}

// ------------------------------------------------------------------------------------------------
// McCDR.estimation (noise_estimation/mccdr.py:122-177 + coherence/BinauralEnhancement.py:24-62), mic pair (1, 2):
// in0 = y complex [B][T][K][M], in1 = Fn [K] (diffuse coherence of the pair); out0 = Gamma [B][T][K].
// state floats: Pxii_1, Pxii_2, Re/Im Pxij_12, MCRA(5)
// ------------------------------------------------------------------------------------------------
DS_HD void op_mccdr(const OpCtx& p, int b, int k) {
    const int M = p.M;
    float p1 = st_at(p, b, 0, k), p2 = st_at(p, b, 1, k);
    cf x12 = mk(st_at(p, b, 2, k), st_at(p, b, 3, k));
    float mc[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) mc[f] = st_at(p, b, 4 + f, k);
    int frm = p.frm_cnt, ell = p.ell;
    const float Fn = p.in1[k];
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K;
        const long long base = (fb + k) * M;
        const cf y0 = mk(p.in0[2 * base], p.in0[2 * base + 1]);
        const cf y1 = mk(p.in0[2 * (base + 1)], p.in0[2 * (base + 1) + 1]);
        const cf y2 = mk(p.in0[2 * (base + 2)], p.in0[2 * (base + 2) + 1]);
        const bool reset = mcra_tick(frm, ell, p.L);
        const float pw0 = cabs2(y0);
        float ym = 0.0f, yp = 0.0f;
        if (k > 0) { const long long q = (fb + k - 1) * M; ym = cabs2(mk(p.in0[2 * q], p.in0[2 * q + 1])); }
        if (k < p.K - 1) { const long long q = (fb + k + 1) * M; yp = cabs2(mk(p.in0[2 * q], p.in0[2 * q + 1])); }
        const float g = mccdr_frame(p1, p2, x12, mc, y1, y2, Fn, ym, pw0, yp, k, p.K, frm, reset, p.L);
        frm += 1; ell += 1;
        p.out0[fb + k] = g;
    }
    st_at(p, b, 0, k) = p1; st_at(p, b, 1, k) = p2; st_at(p, b, 2, k) = x12.x; st_at(p, b, 3, k) = x12.y;
#pragma unroll
    for (int f = 0; f < 5; ++f) st_at(p, b, 4 + f, k) = mc[f];
}

// ------------------------------------------------------------------------------------------------
// McSpp.estimation (noise_estimation/mcspp.py:244-305) with the notebook's online MVDR (example/mvdr.ipynb cell 4:
// steering(Phi_xx) -> compute_mvdr_weight(steer, Phi_vv_inv) -> sum conj(w) y) fused behind it:
// in0 = y complex [B][T][K][M], in1 = Gamma [B][T][K] (op_mccdr);
// out0 = p [B][T][K], out1 = w_pmwf complex [B][T][K][M], out2 = Yout complex [B][T][K] (optional),
// out3 = Phi_xx, out4 = Phi_vv_inv complex [B][T][K][M][M] (optional).
// state floats: rows 0 .. 8 McCDR's, 9 .. 11 unused (the matrices start on a float4 plane, so that their 16-byte groups are known at compile
// time: MCSPP_ROW0), then Phi_yy, Phi_vv Hermitian packed (M*M each), xi, gamma, p
// ------------------------------------------------------------------------------------------------
constexpr int MCSPP_ROW0 = 12;
DS_HD int mcspp_nf(int M) { return MCSPP_ROW0 + 2 * M * M + 3; }

// np.mean(q[fmin:fmax]) of one frame (mcspp.py:260), q = 1 - Gamma; summed in bin order
DS_HD float mcspp_qavg(const float* gamma_frame, int fmin, int fmax) {
    float qsum = 0.0f;
    for (int j = fmin; j < fmax; ++j) qsum += 1.0f - gamma_frame[j];
    return qsum / (float)(fmax - fmin);
}

template <int M> DS_HD void op_mcspp(const OpCtx& p, int b, int k) {
    constexpr int NO = M * (M - 1) / 2;
    constexpr int o0 = MCSPP_ROW0;                                                 // state row offset of the McSpp part
    // rows o0 .. o0 + 2 M M - 1 = Phi_yy (diagonal, upper triangle), Phi_vv (the same), then xi, gamma, p: one register block, moved as
    // whole float4 groups (o0 is a multiple of 4)
    float mat[2 * M * M + 4];
    constexpr int YD = 0, VD = M * M;                                              // Phi_yy / Phi_vv: M diagonal words, then the upper triangle
    st_load_span<o0, 2 * M * M>(p, b, k, mat);
    auto S = [&](int f) -> float { return mat[f]; };
    // entry (i, j) of the Hermitian-packed matrix whose words start at `base` (herm_get's reading of the same words)
    auto hg = [&](int base, int i, int j) -> cf {
        if (i == j) return mk(S(base + i), 0.0f);
        const int w = base + M + 2 * off_index(i < j ? i : j, i < j ? j : i, M);
        const cf v = mk(S(w), S(w + 1));
        return i < j ? v : mk(v.x, -v.y);
    };
    // Hermitian rank-one update of that matrix
    auto rank1 = [&](int base, const cf* Z, float a_, float b_) {
        herm_rank1<M>(mat + base, mat + base + M, Z, a_, b_);
    };
    int frm = p.frm_cnt;
    const int fmin = (int)(500.0 * (2 * (p.K - 1)) / 16000.0), fmax = (int)(2000.0 * (2 * (p.K - 1)) / 16000.0);   // :258-259
    float xi = 0, gam = 0, pp = 0;
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K;
        const long long base = (fb + k) * M;
        cf Z[M];
#pragma unroll
        for (int m = 0; m < M; ++m) Z[m] = mk(p.in0[2 * (base + m)], p.in0[2 * (base + m) + 1]);
        float q = 1.0f - p.in1[fb + k];                                            // compute_q :113-116
        float q_avg;
        if (p.in2) {
            q_avg = p.in2[(long long)(b - p.in2_b0) * p.T + t];                    // mcspp_qavg() ran once per (utterance, frame)
        } else {
            q_avg = mcspp_qavg(p.in1 + fb, fmin, fmax);
        }
        const float dv = fma_(q_avg, 1e-1f, (1.0f - q_avg) * 1e-4f);               // :254-262
        rank1(YD, Z, 0.92f, (float)(1.0 - 0.92));                                  // :264-266
        if (frm < 10) {                                                            // :273-275
#pragma unroll
            for (int f = 0; f < M * M; ++f) mat[VD + f] = mat[YD + f];
            q = 0.99f;
        }
        // estimation_core :201-242 — in double like the reference's complex128 (ds_linalg64.hpp): inv(Phi_vv + dv I) of nearly rank-one
        // matrices and the cancellation Phi_yy - Phi_vv are conditioning-limited; the carried state stays fp32.
        // Register discipline (round 5; the 6-microphone kernel had sat at 512 registers + 260 B of scratch): no double copy of a state
        // matrix is held across a phase — Phi_yy and Phi_xx = Phi_yy - Phi_vv are re-formed from the fp32 state where they are used (the
        // same conversions and the same subtraction: the same values), the eigenvector of the notebook's steering() is taken in front of
        // the LAST estimation_core, while nothing else of size is live (only the 2 M doubles of the vector cross the core), and the
        // noise update comes after every reader of the frame's Phi_vv.  Same operations on the same operands as before: same results
        // (between two phases the fp32 state passes through empty asm statements: otherwise the compiler merges the identical conversions of
        // different phases and keeps the double copies live across them — the very thing this arrangement is there to avoid)
        auto phase_fence = [&]() {
#pragma unroll
            for (int f = 0; f < 2 * M * M; ++f) DS_PIN(mat[f]);
        };
        auto pyy = [&](int i, int j) { return to_cd(hg(YD, i, j)); };
        auto pvv = [&](int i, int j) { return to_cd(hg(VD, i, j)); };
        auto pxx = [&](int i, int j) { return cdsub(to_cd(hg(YD, i, j)), to_cd(hg(VD, i, j))); };   // Phi_xx = Phi_yy - Phi_vv (:212; exact in double)
        auto zd = [&](int m) { return to_cd(Z[m]); };
        // (round 6) inv(Phi_vv + dv I) is kept as the inverse of its Cholesky factor (CholInvD: A^-1 = Li^H Li) — the trace, A^-1 y, the PMWF column
        // and the MVDR weights are products with it; the explicit inverse (2 M^2 doubles) was the operator's register peak
        CholInvD<M> ci;
        cd sv[M];
        double xid = 0.0;
        const int passes = p.repeat ? 2 : 1;
        for (int pass = 0; pass < passes; ++pass) {
        if (pass == 1) {                                                           // update_noise_psd (alpha_d = 0.92) on the fp32 state, then the
            const float at1 = fma_((float)(1.0 - 0.92), pp, 0.92f);                // second estimation_core of repeat=True (:280-282)
            rank1(VD, Z, at1, 1.0f - at1);
        }
        if (p.out2 && pass == passes - 1) {                                        // mvdr.ipynb cell 4: steer_vector = steering(noise_estimator.Phi_xx)
            // the ONE eigenvector steering() keeps (beamformer.py:24), by the direct solve of ds_linalg64.hpp
            // |Phi_xx_ij|^2 for the solver's scale from the fp32 words (a magnitude is all it is used for)
            auto pxx_mag2 = [&](int i, int j) { const cf y_ = hg(YD, i, j), v_ = hg(VD, i, j); const float re = y_.x - v_.x, im = y_.y - v_.y; return fma_(re, re, im * im); };
#if defined(DS_LAGUERRE_COUNT)
            herm_principal_direct_get_d<M>(pxx, sv, DS_LAGUERRE_COUNT, phase_fence, pxx_mag2);
#else
            herm_principal_direct_get_d<M>(pxx, sv, nullptr, phase_fence, pxx_mag2);
#endif
            DS_SCHED_FENCE();
            phase_fence();
        }
        const double dvd = (double)dv;
        ci.factor_invert([&](int i, int j) { cd t_ = pvv(i, j); if (i == j) t_.x += dvd; return t_; });   // :214
        phase_fence();
        double tr = ci.trace_with(pyy);                                            // Re tr(Phi_vv_inv Phi_yy) :217
        phase_fence();
        if (tr - (double)M < 0.0) {                                                // :219-228
            const double dl = frm < 5 ? dvd : 0.0;
            ci.factor_invert([&](int i, int j) { cd t_ = pyy(i, j); if (i == j) t_.x += dl; return t_; });
            // (round 6) with Phi_vv_inv = inv(Phi_yy + dl I) the trace is sum lambda_i / (lambda_i + dl) <= M: xi = tr - M is zero or negative up to
            // rounding (1e-15) and takes its floor — the second trace (M^2 (M + 1) / 2 complex products) computed nothing else
            tr = (double)M;
        }
        phase_fence();
        xid = dmin_(dmax_(tr - (double)M, 1e-6), 1e8);                  // :230
        cd u[M], v[M];
        {
            cd Zd[M];
#pragma unroll
            for (int m = 0; m < M; ++m) Zd[m] = zd(m);
            ci.lower(Zd, u);                                                       // v = Phi_vv_inv y; y^H v = |Li y|^2
        }
        double yv = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) yv += cdabs2(u[i]);
        ci.upper(u, v);
        // v^H Phi_yy v of a Hermitian matrix by its triangle: sum_i Pyy_ii |v_i|^2 + 2 Re sum_{i<j} conj(v_i) Pyy_ij v_j
        double vPv = 0.0, vPo = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            vPv = fmad_(pyy(i, i).x, cdabs2(v[i]), vPv);
#pragma unroll
            for (int j = i + 1; j < M; ++j) {
                const cd c_ = cdmulc(v[j], v[i]), y_ = pyy(i, j);                    // conj(v_i) v_j
                vPo = fmad_(c_.x, y_.x, fmad_(-c_.y, y_.y, vPo));                    // Re(conj(v_i) v_j Pyy_ij)
            }
        }
        vPv = fmad_(2.0, vPo, vPv);
        const double gamd = dmin_(dmax_(vPv - yv, 1e-6), 1e8);                       // :232-236
        const double qd = (double)q;
        const double r1x = rcp_fast_d(1.0 + xid);
        // compute_p :75-92 (reciprocals: seed + two Newton steps).  e^(-gamma / (1 + xi)) from the fp32 hardware exponential: its 1e-7 of relative
        // error is 1e-7 of p at most — the double-precision library routine was 80 instructions and a dozen hoisted constants for nothing
        const double egx = (double)exp2_(-(float)(gamd * r1x) * 1.44269504088896341f);
        double ppd = rcp_fast_d(1.0 + qd * rcp_fast_d(1.0 - qd) * (1.0 + xid) * egx);
        ppd = dmin_(dmax_(ppd, 0.0), 1.0);
        xi = (float)xid; gam = (float)gamd; pp = (float)ppd;
        phase_fence();
        }
        const long long ob = fb + k;
        p.out0[ob] = pp;
        const double wsc = rcp_fast_d(10.0 + xid);                                 // compute_pmwf_weight beta = 10 :283
        if (p.out1) {
            cd x0[M], u[M], w[M];
#pragma unroll
            for (int j = 0; j < M; ++j) x0[j] = pxx(j, 0);
            ci.lower(x0, u); ci.upper(u, w);                                       // Phi_vv_inv Phi_xx[:, 0]
#pragma unroll
            for (int i = 0; i < M; ++i) { p.out1[2 * (ob * M + i)] = (float)(w[i].x * wsc); p.out1[2 * (ob * M + i) + 1] = (float)(w[i].y * wsc); }
        }
        if (p.out3) {
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    const long long q2 = 2 * ((ob * M + i) * M + j);
                    const cd x_ = pxx(i, j), iv_ = ci.inverse_entry(i, j);
                    p.out3[q2] = (float)x_.x; p.out3[q2 + 1] = (float)x_.y;
                    p.out4[q2] = (float)iv_.x; p.out4[q2 + 1] = (float)iv_.y;
                }
        }
        if (p.out2) {                                                              // w = compute_mvdr_weight(steer_vector, Phi_vv_inv); Yout = w^H y
            cd u[M], w[M];                                                          // w = A^-1 a / (a^H A^-1 a), a^H A^-1 a = |Li a|^2 (beamformer.py:133-155)
            ci.lower(sv, u);
            double den = 0.0;
#pragma unroll
            for (int m = 0; m < M; ++m) den += cdabs2(u[m]);
            ci.upper(u, w);
            cd Y = mkd(0.0, 0.0);
#pragma unroll
            for (int m = 0; m < M; ++m) Y = cdfmac(Y, zd(m), w[m]);
            const double rden = rcp_fast_d(den);
            p.out2[2 * ob] = (float)(Y.x * rden); p.out2[2 * ob + 1] = (float)(Y.y * rden);
        }
        if (!p.repeat) {                                                           // update_noise_psd (alpha_d = 0.92) on the fp32 state: behind
            const float at = fma_((float)(1.0 - 0.92), pp, 0.92f);                 // every reader of this frame's Phi_vv
            rank1(VD, Z, at, 1.0f - at);
        }
        frm += 1;
    }
    if (p.T > 0) {
        mat[2 * M * M] = xi; mat[2 * M * M + 1] = gam; mat[2 * M * M + 2] = pp;
        st_store_span<o0, 2 * M * M + 3>(p, b, k, mat);
    } else {
        st_store_span<o0, 2 * M * M>(p, b, k, mat);
    }
}

// McSpp without the notebook-MVDR / matrix outputs (OP_MCSPP_LEAN, the SubbandGSC chain): the same estimation_core, but nothing here
// needs inv(Phi_vv + dv I) as a matrix — tr(A^-1 Phi_yy), A^-1 y and the PMWF column come from the inverse of the Cholesky factor of the
// Hermitian-packed state (Chol::invert / trace_with / apply), which is less work and fewer live registers than forming the explicit
// inverse, and less than a pair of substitutions per column of Phi_yy (the first form of this operator: 233 registers, 211 now at M = 6).
// where the fused operator's blocking filters put their error spectra, complex [B * M][T][K]: on the device ONE buffer descriptor for the
// array, the lane's (utterance, bin) as the 32-bit offset and (filter, frame) in the scalar offset — M per-filter 64-bit pointers walked
// frame by frame were twelve more registers than the operator's two-waves-per-SIMD budget has (the launcher keeps the array under 4 GB)
#if defined(__HIP_DEVICE_COMPILE__)
struct FanErrOut {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;
    int T, K;
    __device__ FanErrOut(const OpParams& p, int b, int k, int M) : T(p.T), K(p.K) {
        const long long bytes = (long long)p.B * M * p.T * p.K * 8;
        rs = __builtin_amdgcn_make_buffer_rsrc(p.fan_e, 0, (int)(unsigned)(bytes > 0xffffffffLL ? 0xffffffffLL : bytes), 0x00020000);
        voff = (unsigned)((((long long)b * M * p.T) * p.K + k) * 8);
    }
    __device__ void put(int m, int t, cf e) const {
        const unsigned soff = (unsigned)((m * T + t) * K * 8);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e.x), rs, voff, soff, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e.y), rs, voff + 4, soff, 0);
    }
};
#else
struct FanErrOut {
    float* base; int T, K;
    FanErrOut(const OpParams& p, int b, int k, int M) : base(p.fan_e ? p.fan_e + 2 * ((((long long)b * M * p.T) * p.K) + k) : nullptr), T(p.T), K(p.K) {}
    void put(int m, int t, cf e) const { float* q = base + 2 * (long long)(m * T + t) * K; q[0] = e.x; q[1] = e.y; }
};
#endif
// FAN (with STEADY; OP_MCSPP_STEADY_FAN): the M RLS blocking filters of the utterance (RlsFan, the program of op_subrls_fan) run on the same
// frame in the same thread — the frame's M spectra are their desired signals.  Their 36 state floats stay within the 256 registers of the
// two-waves-per-SIMD budget the operator lives in anyway
template <int M, bool STEADY = false, bool FAN = false> DS_HD void op_mcspp_lean(const OpCtx& p, int b, int k) {
    constexpr int NO = M * (M - 1) / 2;
    constexpr int o0 = MCSPP_ROW0;
    RlsFan<2, FAN ? M : 1> fan;
    const OpCtx* fc = nullptr;
    const FanErrOut eout(p, b, k, M);
    float* const fpark = (FAN && p.spill) ? p.spill + (M * M + 1) * p.spill_stride : nullptr;      // behind Phi_vv's parking rows
    if constexpr (FAN) {
        fc = static_cast<const OpCtx*>(p.fan_ctx);
        fan.load(*fc, b * M, k);
        if (fpark) { fan.park(fpark, p.spill_stride); DS_COMPILER_FENCE(); }
    }
    // rows o0 .. o0 + 2 M M - 1 = Phi_yy (diagonal, upper triangle), Phi_vv (the same), then xi, gamma, p: one register block, moved as
    // whole float4 groups (o0 is a multiple of 4)
    float mat[2 * M * M + 4];
    float* yd = mat;
    float* yo = mat + M;
    float* vd = mat + M * M;
    float* vo = mat + M * M + M;
    st_load_span<o0, 2 * M * M>(p, b, k, mat);
    int frm = p.frm_cnt;
    const int fmin = (int)(500.0 * (2 * (p.K - 1)) / 16000.0), fmax = (int)(2000.0 * (2 * (p.K - 1)) / 16000.0);   // :258-259
    float xi = 0, gam = 0, pp = 0;
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K;
        const long long base = (fb + k) * M;
        cf Z[M];
#pragma unroll
        for (int m = 0; m < M; ++m) Z[m] = mk(p.in0[2 * (base + m)], p.in0[2 * (base + m) + 1]);
        if constexpr (FAN) {                                                       // SubbandGSC.py:217-223: bm[m].update(fixed, aligned[m]) for every m
            cf err[M];
            if (fpark) { DS_COMPILER_FENCE(); fan.unpark(fpark, p.spill_stride); }
            fan.step(mk(p.fan_x[2 * (fb + k)], p.fan_x[2 * (fb + k) + 1]), Z, p.fan_lam, p.fan_mu, err);
            if (fpark) { fan.park(fpark, p.spill_stride); DS_COMPILER_FENCE(); }
#pragma unroll
            for (int m = 0; m < M; ++m) eout.put(m, t, err[m]);
        }
        float q = 1.0f - p.in1[fb + k];                                            // compute_q :113-116
        const float q_avg = p.in2 ? p.in2[(long long)(b - p.in2_b0) * p.T + t] : mcspp_qavg(p.in1 + fb, fmin, fmax);
        const float dv = fma_(q_avg, 1e-1f, (1.0f - q_avg) * 1e-4f);               // :254-262
        herm_rank1<M>(yd, yo, Z, 0.92f, (float)(1.0 - 0.92));                      // :264-266
        if (frm < 10) {                                                            // :273-275
#pragma unroll
            for (int f = 0; f < M; ++f) vd[f] = yd[f];
#pragma unroll
            for (int f = 0; f < 2 * NO; ++f) vo[f] = yo[f];
            q = 0.99f;
        }
        // estimation_core :201-242 with A = Phi_vv + dv I = L L^H, through L^-1 (Chol::invert): Re tr(A^-1 Phi_yy) is the sum of the quadratic
        // forms of Phi_yy in the rows of L^-1 and A^-1 y two triangular products — one triangular inverse instead of a substitution pair per
        // column of Phi_yy, and no right-hand side or solution vector live next to the factor (the six solves held the kernel at two waves
        // per SIMD and were half of its arithmetic)
        Chol<M> ch;
        ch.factor(vd, vo, dv);
        // Phi_vv is not needed again before the noise update at the end of the frame: parked (LDS) while the rest of the frame runs
        if (p.spill) {
#pragma unroll
            for (int f = 0; f < M; ++f) p.spill[f * p.spill_stride] = vd[f];
#pragma unroll
            for (int f = 0; f < 2 * NO; ++f) p.spill[(M + f) * p.spill_stride] = vo[f];
            DS_COMPILER_FENCE();
        }
        ch.invert();
        DS_SCHED_FENCE();
        float tr = ch.trace_with(yd, yo);
        DS_SCHED_FENCE();
        // :219-228: where xi < 0 the reference falls back to A = Phi_yy (+ dv I in the first five frames).  From frame 5 on that makes
        // A^-1 Phi_yy the identity: tr - M and y^H A^-1 Phi_yy A^-1 y - y^H A^-1 y are zero up to rounding (1e-14 in the reference's
        // doubles), so both clamp to their floor 1e-6 (:230,236) — taken here as exactly that, without a second factorisation and trace
        // whose fp32 rounding would only land somewhere in 1e-6 .. 1e-5.  The PMWF weights (out1) still need A^-1.
        // STEADY: the variant for calls that start at frame 5 or later and do not ask for the PMWF weights — the second factorisation does
        // not exist in it, which is what brings the 6-microphone kernel under 256 registers (two waves per SIMD)
        const bool fell = tr - (float)M < 0.0f, ident = fell && (STEADY || frm >= 5);
        if constexpr (!STEADY) {
            if (fell && (!ident || p.out1)) {
                ch.factor(yd, yo, frm < 5 ? dv : 0.0f);
                ch.invert();
                if (!ident) tr = ch.trace_with(yd, yo);
            }
        }
        xi = ident ? 1e-6f : fminf_(fmaxf_(tr - (float)M, 1e-6f), 1e8f);           // :230
        gam = 1e-6f;
        if (!ident) {
            cf v[M];
            ch.apply(Z, v);                                                        // v = A^-1 y
            float yv = 0.0f, vPv = 0.0f;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                yv = fma_(Z[i].x, v[i].x, fma_(Z[i].y, v[i].y, yv));                // Re(conj(y_i) v_i)
                cf acc = mk(0.0f, 0.0f);
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    // (below the diagonal Phi_ij = conj(Phi_ji): acc + conj(Phi_ji) v_j as cfmac(acc, v_j, Phi_ji) — the same products in the same
                    // nesting, without the conjugate built in a register pair)
                    if (j < i) { const int w = off_index(j, i, M); acc = cfmac(acc, v[j], mk(yo[2 * w], yo[2 * w + 1])); }
                    else acc = cfma(acc, herm_get<M>(yd, yo, i, j), v[j]);
                }
                vPv = fma_(v[i].x, acc.x, fma_(v[i].y, acc.y, vPv));                // Re(conj(v_i) (Phi_yy v)_i)
            }
            gam = fminf_(fmaxf_(vPv - yv, 1e-6f), 1e8f);                           // :232-236
        }
        pp = rcp_(1.0f + div_(q, 1.0f - q) * (1.0f + xi) * exp_(-1.0f * div_(gam, 1.0f + xi)));   // compute_p :75-92
        pp = fminf_(fmaxf_(pp, 0.0f), 1.0f);
        const long long ob = fb + k;
        if (p.spill) {
            DS_COMPILER_FENCE();
#pragma unroll
            for (int f = 0; f < M; ++f) vd[f] = p.spill[f * p.spill_stride];
#pragma unroll
            for (int f = 0; f < 2 * NO; ++f) vo[f] = p.spill[(M + f) * p.spill_stride];
        }
        if constexpr (!STEADY) {
            if (p.out1) {                                                          // compute_pmwf_weight beta = 10 :283, Phi_xx before the noise update
                const float wsc = 1.0f / (10.0f + xi);
                cf col[M], w[M];
#pragma unroll
                for (int i = 0; i < M; ++i) col[i] = csub(herm_get<M>(yd, yo, i, 0), herm_get<M>(vd, vo, i, 0));
                ch.apply(col, w);
#pragma unroll
                for (int i = 0; i < M; ++i) { p.out1[2 * (ob * M + i)] = w[i].x * wsc; p.out1[2 * (ob * M + i) + 1] = w[i].y * wsc; }
            }
        }
        const float at = fma_((float)(1.0 - 0.92), pp, 0.92f);                     // update_noise_psd (alpha_d = 0.92)
        herm_rank1<M>(vd, vo, Z, at, 1.0f - at);
        p.out0[ob] = pp;
        frm += 1;
    }
    if constexpr (FAN) {
        if (fpark) { DS_COMPILER_FENCE(); fan.unpark(fpark, p.spill_stride); }
        fan.store(*fc, b * M, k);
    }
    if (p.T > 0) {
        mat[2 * M * M] = xi; mat[2 * M * M + 1] = gam; mat[2 * M * M + 2] = pp;
        st_store_span<o0, 2 * M * M + 3>(p, b, k, mat);
    } else {
        st_store_span<o0, 2 * M * M>(p, b, k, mat);
    }
}

// stateless: steering(XXs) — in0 = XX complex [B][K][M][M] -> out0 = v complex [B][K][M]
template <int M> DS_HD void op_steering(const OpCtx& p, int b, int k) {
    cd A[M][M], v[M];
    const long long base = ((long long)b * p.K + k) * M * M;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) A[i][j] = mkd((double)p.in0[2 * (base + i * M + j)], (double)p.in0[2 * (base + i * M + j) + 1]);
#pragma unroll
    for (int i = 0; i < M; ++i) {                        // eigh reads one triangle: use the lower one like LAPACK 'L'
        A[i][i].y = 0.0;
#pragma unroll
        for (int j = i + 1; j < M; ++j) A[i][j] = cdconj(A[j][i]);
    }
    herm_principal_d<M>(A, v);                           // in double: the eigenvector of a nearly degenerate pair is conditioning-limited
    const long long ob = ((long long)b * p.K + k) * M;
#pragma unroll
    for (int m = 0; m < M; ++m) { p.out0[2 * (ob + m)] = (float)v[m].x; p.out0[2 * (ob + m) + 1] = (float)v[m].y; }
}

// stateless: compute_mvdr_weight — in0 = steer complex [B][K][M], in1 = Rvv_inv complex [B][K][M][M] -> out0 = w [B][K][M]
template <int M> DS_HD void op_mvdrw(const OpCtx& p, int b, int k) {
    cd R[M][M], a[M], w[M];
    const long long rb = ((long long)b * p.K + k) * M * M, ab = ((long long)b * p.K + k) * M;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        a[i] = mkd((double)p.in0[2 * (ab + i)], (double)p.in0[2 * (ab + i) + 1]);
#pragma unroll
        for (int j = 0; j < M; ++j) R[i][j] = mkd((double)p.in1[2 * (rb + i * M + j)], (double)p.in1[2 * (rb + i * M + j) + 1]);
    }
    mvdr_weight_d<M>(R, a, w);
#pragma unroll
    for (int m = 0; m < M; ++m) { p.out0[2 * (ab + m)] = (float)w[m].x; p.out0[2 * (ab + m) + 1] = (float)w[m].y; }
}


// stateless: compute_pmwf_weight(xi, Rxx, Rvv_inv, beta) — beamformer/beamformer.py:100-130: w = (Rvv_inv Rxx)[:, 0] / (beta + xi).
// in0 = xi [B][K], in1 = Rxx complex [B][K][M][M], in2 = Rvv_inv complex [B][K][M][M]; beta in p.mu -> out0 = w complex [B][K][M]
template <int M> DS_HD void op_pmwfw(const OpCtx& p, int b, int k) {
    const long long rb = ((long long)b * p.K + k) * M * M, ob = ((long long)b * p.K + k) * M;
    cd x0[M];
#pragma unroll
    for (int i = 0; i < M; ++i) x0[i] = mkd((double)p.in1[2 * (rb + i * M)], (double)p.in1[2 * (rb + i * M) + 1]);     // column 0 of Rxx
    const double den = 1.0 / ((double)p.mu + (double)p.in0[(long long)b * p.K + k]);
#pragma unroll
    for (int i = 0; i < M; ++i) {
        cd t = mkd(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) t = cdfma(t, mkd((double)p.in2[2 * (rb + i * M + j)], (double)p.in2[2 * (rb + i * M + j) + 1]), x0[j]);
        p.out0[2 * (ob + i)] = (float)(t.x * den); p.out0[2 * (ob + i) + 1] = (float)(t.y * den);
    }
}

// stateless: get_gev_vector(target_psd_matrix, noise_psd_matrix) — beamformer/beamformer.py:79-97.  in0 = target complex [B][K][M][M],
// in1 = noise complex [B][K][M][M] -> out0 = v complex [B][K][M] (principal generalised eigenvector, v^H N v = 1; phase: ds_linalg64.hpp)
template <int M> DS_HD void op_gev(const OpCtx& p, int b, int k) {
    cd A[M][M], N[M][M], v[M];
    const long long base = ((long long)b * p.K + k) * M * M, ob = ((long long)b * p.K + k) * M;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
            A[i][j] = mkd((double)p.in0[2 * (base + i * M + j)], (double)p.in0[2 * (base + i * M + j) + 1]);
            N[i][j] = mkd((double)p.in1[2 * (base + i * M + j)], (double)p.in1[2 * (base + i * M + j) + 1]);
        }
#pragma unroll
    for (int i = 0; i < M; ++i) {                        // scipy's eigh reads the lower triangles
        A[i][i].y = 0.0; N[i][i].y = 0.0;
#pragma unroll
        for (int j = i + 1; j < M; ++j) { A[i][j] = cdconj(A[j][i]); N[i][j] = cdconj(N[j][i]); }
    }
    if (!herm_gev_principal_d<M>(A, N, v)) {             // :94-96: ones / trace(noise) * sensors — the COMPLEX trace of the matrix as given
        double trx = 0.0, try_ = 0.0;                    // (its diagonal's imaginary parts as they came in: np.trace does not symmetrise)
#pragma unroll
        for (int i = 0; i < M; ++i) { trx += (double)p.in1[2 * (base + i * M + i)]; try_ += (double)p.in1[2 * (base + i * M + i) + 1]; }
        const double d = trx * trx + try_ * try_;
#pragma unroll
        for (int i = 0; i < M; ++i) v[i] = mkd((double)M * trx / d, -(double)M * try_ / d);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) { p.out0[2 * (ob + m)] = (float)v[m].x; p.out0[2 * (ob + m) + 1] = (float)v[m].y; }
}

// stateless: blind_analytic_normalization(vector, noise_psd_matrix, eps) — beamformer/beamformer.py:34-63:
// w * |sqrt(w^H N N w)| / (|w^H N w| + eps).  in0 = vector complex [B][K][M], in1 = noise complex [B][K][M][M]; eps in p.reg -> out0
template <int M> DS_HD void op_ban(const OpCtx& p, int b, int k) {
    const long long rb = ((long long)b * p.K + k) * M * M, ob = ((long long)b * p.K + k) * M;
    cd w[M], Nw[M], NNw[M];
#pragma unroll
    for (int i = 0; i < M; ++i) w[i] = mkd((double)p.in0[2 * (ob + i)], (double)p.in0[2 * (ob + i) + 1]);
    auto Nat = [&](int i, int j) { return mkd((double)p.in1[2 * (rb + i * M + j)], (double)p.in1[2 * (rb + i * M + j) + 1]); };
#pragma unroll
    for (int i = 0; i < M; ++i) {
        cd t = mkd(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) t = cdfma(t, Nat(i, j), w[j]);
        Nw[i] = t;
    }
#pragma unroll
    for (int i = 0; i < M; ++i) {
        cd t = mkd(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < M; ++j) t = cdfma(t, Nat(i, j), Nw[j]);
        NNw[i] = t;
    }
    cd nom = mkd(0.0, 0.0), den = mkd(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < M; ++i) { nom = cdfmac(nom, NNw[i], w[i]); den = cdfmac(den, Nw[i], w[i]); }    // sum conj(w_i) (..)_i
    const double scale = sqrt(sqrt(cdabs2(nom))) / (sqrt(cdabs2(den)) + (double)p.reg);                // |sqrt(z)| = sqrt(|z|)
#pragma unroll
    for (int m = 0; m < M; ++m) { p.out0[2 * (ob + m)] = (float)(w[m].x * scale); p.out0[2 * (ob + m) + 1] = (float)(w[m].y * scale); }
}

// stateless: phase_correction(vector) — beamformer/beamformer.py:66-76: bin f is rotated by exp(-j angle(sum_m w[f, m] conj(w[f - 1, m])))
// with w[f - 1] already corrected: a serial walk over the bins of one utterance (the thread of bin 0 does it).  in0 -> out0, complex [B][K][M]
template <int M> DS_HD void op_phasecorr(const OpCtx& p, int b, int k) {
    if (k != 0) return;
    const long long ub = (long long)b * p.K * M;
    cd prev[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        prev[m] = mkd((double)p.in0[2 * (ub + m)], (double)p.in0[2 * (ub + m) + 1]);
        p.out0[2 * (ub + m)] = (float)prev[m].x; p.out0[2 * (ub + m) + 1] = (float)prev[m].y;
    }
    for (int f = 1; f < p.K; ++f) {
        cd cur[M];
        cd s = mkd(0.0, 0.0);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            cur[m] = mkd((double)p.in0[2 * (ub + (long long)f * M + m)], (double)p.in0[2 * (ub + (long long)f * M + m) + 1]);
            s = cdfmac(s, cur[m], prev[m]);
        }
        const double n = sqrt(cdabs2(s));
        const cd rot = n > 0.0 ? cdscale(cdconj(s), 1.0 / n) : mkd(1.0, 0.0);     // exp(-j angle(s)); angle(0) = 0
#pragma unroll
        for (int m = 0; m < M; ++m) {
            prev[m] = cdmul(cur[m], rot);
            p.out0[2 * (ub + (long long)f * M + m)] = (float)prev[m].x; p.out0[2 * (ub + (long long)f * M + m) + 1] = (float)prev[m].y;
        }
    }
}

// dispatch one (b, k) of an operator; OP and (for the matrix operators) M are compile-time so every operator
// gets its own register allocation
// ------------------------------------------------------------------------------------------------
// adaptivebeamfomer frame loop on STFT frames — beamformer/adaptivebeamformer.py:69-120 with getweights
// (beamformer.py:306-336), i.e. the per-bin program of the fused DS_ALGO_ADAPTIVE kernel as a frame-level operator:
// in0 = Z complex [B][T][K][M], in1 = optional post-filter gain [B][T][K] (has_p); out0 = Y complex [B][T][K] (= w^H Z * gain).
// state floats: Rvv Hermitian-packed (M diagonal reals, M(M-1)/2 complex upper entries), MCRA S,Smin,Stmp,p,lambda_d
// ------------------------------------------------------------------------------------------------
template <int M> DS_HD void op_adaptive(const OpCtx& p, int b, int k) {
    typedef StateLayout<M, ALGO_ADAPTIVE, false> SL;
    float st[SL::NF];
    st_load_span<0, SL::NF>(p, b, k, st);
    cf a[M];
    const cf* sv = p.steer + (long long)b * p.steer_batch_stride + (long long)k * M;
#pragma unroll
    for (int m = 0; m < M; ++m) a[m] = sv[m];
    Params q;
    q.method = p.method; q.alpha_y = 0.8f; q.beta_y = (float)(1.0 - 0.8); q.alpha_v = p.alpha_v; q.beta_v = p.beta_v; q.gate = p.gate; q.gate_kinv = 0; q.diag = p.diag; q.diag_floor = p.diag_floor;
    int frm = p.frm_cnt, ell = p.ell;
    for (int t = 0; t < p.T; ++t) {
        const long long fb = ((long long)b * p.T + t) * p.K;
        const long long base = (fb + k) * M;
        cf Z[M];
#pragma unroll
        for (int m = 0; m < M; ++m) Z[m] = mk(p.in0[2 * (base + m)], p.in0[2 * (base + m) + 1]);
        const bool reset = mcra_tick(frm, ell, p.L);
        float ym = 0.0f, yp = 0.0f;
        if (k > 0) { const long long i = (fb + k - 1) * M; ym = cabs2(mk(p.in0[2 * i], p.in0[2 * i + 1])); }
        if (k < p.K - 1) { const long long i = (fb + k + 1) * M; yp = cabs2(mk(p.in0[2 * i], p.in0[2 * i + 1])); }
        mcra_bin(st + SL::MC_S, k, p.K, ym, cabs2(Z[0]), yp, frm, reset, p.L);          // :81
        frm += 1; ell += 1;
        cf Y = adaptive_bin<M, false>(st, Z, a, q);
        if (p.has_p) Y = cscale(Y, p.in1[fb + k]);
        p.out0[2 * (fb + k)] = Y.x; p.out0[2 * (fb + k) + 1] = Y.y;
    }
    st_store_span<0, SL::NF>(p, b, k, st);
}

template <int OP, int M> DS_HD void run_op_t(const OpCtx& p, int b, int k) {
    if constexpr (OP == OP_MCRA) op_mcra(p, b, k);
    else if constexpr (OP == OP_OMLSA) op_omlsa(p, b, k);
    else if constexpr (OP == OP_SUBLMS) op_sublms(p, b, k);
    else if constexpr (OP == OP_SUBRLS) op_subrls(p, b, k);
    else if constexpr (OP == OP_MCCDR) op_mccdr(p, b, k);
    else if constexpr (OP == OP_MCMCRA) op_mcmcra<M>(p, b, k);
    else if constexpr (OP == OP_MCSPPBASE) op_mcsppbase<M>(p, b, k);
    else if constexpr (OP == OP_MCSPP) op_mcspp<M>(p, b, k);
    else if constexpr (OP == OP_MCSPP_LEAN) op_mcspp_lean<M>(p, b, k);
    else if constexpr (OP == OP_MCSPP_STEADY) op_mcspp_lean<M, true>(p, b, k);
    else if constexpr (OP == OP_MCSPP_STEADY_FAN) op_mcspp_lean<M, true, true>(p, b, k);
    else if constexpr (OP == OP_STEERING) op_steering<M>(p, b, k);
    else if constexpr (OP == OP_MVDRW) op_mvdrw<M>(p, b, k);
    else if constexpr (OP == OP_PMWFW) op_pmwfw<M>(p, b, k);
    else if constexpr (OP == OP_GEV) op_gev<M>(p, b, k);
    else if constexpr (OP == OP_BAN) op_ban<M>(p, b, k);
    else if constexpr (OP == OP_PHASECORR) op_phasecorr<M>(p, b, k);
    else if constexpr (OP == OP_ADAPTIVE) op_adaptive<M>(p, b, k);
}

inline bool op_is_linalg(int op) { return op == OP_STEERING || op == OP_MVDRW || op == OP_PMWFW || op == OP_GEV || op == OP_BAN || op == OP_PHASECORR; }
inline bool op_is_matrix(int op) { return op == OP_MCMCRA || op == OP_MCSPPBASE || op == OP_MCSPP || op == OP_ADAPTIVE || op == OP_MCSPP_LEAN || op == OP_MCSPP_STEADY || op == OP_MCSPP_STEADY_FAN || op_is_linalg(op); }

// is (op, M) a supported combination?  (matrix operators: M in {2, 4, 6, 8}; McSpp / steering / mvdr weight: {2, 4, 6})
inline bool op_supported(int op, int M) {
    if (op == OP_MCMCRA || op == OP_MCSPPBASE || op == OP_ADAPTIVE) return M == 2 || M == 4 || M == 6 || M == 8;
    if (op == OP_MCSPP || op == OP_MCSPP_LEAN || op == OP_MCSPP_STEADY || op == OP_MCSPP_STEADY_FAN) return M == 2 || M == 4 || M == 6;
    if (op_is_linalg(op)) return M >= 2 && M <= 6;          // the stateless helpers also for the 3- and 5-microphone arrays the frame kernels take
    return true;
}

#define DS_OP_M_LIST(X, OP_) X(OP_, 2) X(OP_, 4) X(OP_, 6) X(OP_, 8)
#define DS_OP_M3_LIST(X, OP_) X(OP_, 2) X(OP_, 4) X(OP_, 6)
#define DS_OP_ML_LIST(X, OP_) X(OP_, 2) X(OP_, 3) X(OP_, 4) X(OP_, 5) X(OP_, 6)
// shelved experiments (make SHELVED=1; the CPU emulator always has them): built, bit-identical to what ships, measured no faster
#if defined(DS_WITH_SHELVED) || !defined(__HIPCC__)
#define DS_OP_SHELVED_LIST(X) DS_OP_M3_LIST(X, OP_MCSPP_STEADY_FAN)
#else
#define DS_OP_SHELVED_LIST(X)
#endif
#if !defined(DS_FOR_EACH_OP)   /* (a scratch build may name the few operators it wants: scripts/asm_ops.sh) */
#define DS_FOR_EACH_OP(X) \
    X(OP_MCRA, 1) X(OP_OMLSA, 1) X(OP_SUBLMS, 1) X(OP_SUBRLS, 1) X(OP_MCCDR, 1) \
    DS_OP_M_LIST(X, OP_MCMCRA) DS_OP_M_LIST(X, OP_MCSPPBASE) DS_OP_M_LIST(X, OP_ADAPTIVE) \
    DS_OP_M3_LIST(X, OP_MCSPP) DS_OP_M3_LIST(X, OP_MCSPP_LEAN) DS_OP_M3_LIST(X, OP_MCSPP_STEADY) DS_OP_SHELVED_LIST(X) DS_OP_ML_LIST(X, OP_STEERING) DS_OP_ML_LIST(X, OP_MVDRW) \
    DS_OP_ML_LIST(X, OP_PMWFW) DS_OP_ML_LIST(X, OP_GEV) DS_OP_ML_LIST(X, OP_BAN) DS_OP_ML_LIST(X, OP_PHASECORR)
#endif

// runtime dispatch for the serial CPU run in tests/emul (the GPU launches one specialised kernel per (OP, M))
inline void run_op(int op, const OpCtx& p, int b, int k) {
#define X(OP_, M_) if (op == OP_ && (!op_is_matrix(OP_) || p.M == M_)) { run_op_t<OP_, M_>(p, b, k); return; }
    DS_FOR_EACH_OP(X)
#undef X
}

// ------------------------------------------------------------------------------------------------
// Time-domain front-end conditioning either side of the STFT:
//   FilterDcNotch16.filter_dc_notch16   adaptivefilter/feature.py:32-49   (2-state IIR, one lane per (utterance, channel))
//   TimeAlignment.process / fir_filter  beamformer/fixedbeamformer.py:13-93 (fractional-delay FIR bank with carried history)
// ------------------------------------------------------------------------------------------------
struct TdParams {
    int B, M, n, L;            // utterances, channels, samples in this call, FIR taps
    const float* x;            // notch: [B][M][n] ; FIR: [B][n][M]
    float* y;                  // same layout as x
    float* mean;               // FIR: optional [B][n] channel mean of y (SubbandGSC.fixed_beamformer, SubbandGSC.py:143)
    float* diff;               // FIR: optional [B][n][M-1] adjacent-pair differences y[m] - y[m+1] (TDGSC.blocking_matrix, TDGSC.py:69-87)
    int y_chan_major;          // FIR: write y as [B][M][n] (what the per-channel transforms of the SubbandGSC chain read) instead of [B][n][M]
    int x_chan_major;          // FIR: read x as [B][M][n]
    long long x_bstride, x_cstride;   // notch: element strides of x between utterances / channels (0 = dense [B][M][n])
    double* mem;               // notch: [B][M][2] doubles (the recursion and its carried memory run in double, see td_dcnotch)
    const float* coef;         // FIR: [L][M]
    const float* cache_in;     // FIR: [B][M][L-1], channel-major: a channel's history is one contiguous row.  (Interleaved [L-1][M] rows made
                               // every channel pass of the kernel a stride-M partial write of the same lines: 1.3x the history in extra HBM
                               // reads AND writes, profiles/r03e/cfg5_stage_budget.md)
    float* cache_out;          // FIR: [B][M][L-1] (the other half of a ping-pong pair)
    const int* dev_parity;     // FIR: optional device-resident call parity (dev_cnt[3] of the handle): odd = the two halves swap roles
    double radius;             // notch: pole radius as the decimal the caller wrote (decimal_double: 0.98f -> 0.98, the reference's Python float)
};

// FilterDcNotch16.filter_dc_notch16 (adaptivefilter/feature.py:32-49) in DOUBLE, memory included (round 5).  The recursion has its poles at
// radius 0.98: run in fp32 it put 1.4e-5 of relative error on its output (the FIR bank behind it: 1.4e-7), and the chain's speech-presence
// estimator amplifies exactly that — on a recording at ten times its level, where McSpp's dv I loading no longer damps it, the fp64
// REFERENCE estimator itself moves p by 4e-3 under a 1e-5 input perturbation (scratch/g22_level2.py) and the blocking filters integrate
// it: 1.2e-4 of the output on the G22 fixtures, 3e-5 with the recursion in double (scratch/g22_level.py; at recording level 4e-6 -> 1e-6).
// Four double operations per sample on the one lane per row that runs the recursion anyway (the kernel is bound by that lane's dependent
// chain, not by the vector rate); samples in and out stay fp32.  (Tried beside it and dropped: xi and gamma of the chain's McSpp in their
// cancellation-free forms tr(A^-1 (Phi_yy - A)) and v^H (Phi_yy - A) v — no measurable change in p, 32 registers more.)
DS_HD double notch_den2(double r) { return r * r + 0.7 * (1.0 - r) * (1.0 - r); }
DS_HD float notch_step(double& m0, double& m1, double r, double den2, float vin_) {
    const double vin = (double)vin_;
    const double vout = m0 + vin;
    m0 = m1 + 2.0 * (-vin + r * vout);
    m1 = vin - den2 * vout;
    return (float)(r * vout);
}
DS_HD void td_dcnotch(const TdParams& p, int b, int m) {
    const double r = p.radius, den2 = notch_den2(p.radius);
    double m0 = p.mem[((long long)b * p.M + m) * 2], m1 = p.mem[((long long)b * p.M + m) * 2 + 1];
    const float* x = p.x_bstride ? p.x + (long long)b * p.x_bstride + (long long)m * p.x_cstride : p.x + ((long long)b * p.M + m) * p.n;
    float* y = p.y + ((long long)b * p.M + m) * p.n;
    for (int i = 0; i < p.n; ++i) y[i] = notch_step(m0, m1, r, den2, x[i]);
    p.mem[((long long)b * p.M + m) * 2] = m0; p.mem[((long long)b * p.M + m) * 2 + 1] = m1;
}

// one output sample (all channels) of the FIR bank: y[n][m] = sum_i c[i][m] x[n - i][m], history from the cache
DS_HD void td_fir(const TdParams& p, int b, int i) {
    const int M = p.M, L = p.L;
    const float* x = p.x + (long long)b * p.n * M;
    const long long xs = p.x_chan_major ? 1 : M, xc = p.x_chan_major ? p.n : 1;      // sample / channel strides of x
    const float* cache = p.cache_in + (long long)b * (L - 1) * M;
    float acc_mean = 0.0f, prev = 0.0f;
    for (int m = 0; m < M; ++m) {
        float acc = 0.0f;
        for (int j = 0; j < L; ++j) {
            const int s = i - j;                                        // sample index relative to this call
            const float v = s >= 0 ? x[(long long)s * xs + m * xc] : cache[(long long)m * (L - 1) + (L - 1 + s)];
            acc = fma_(p.coef[(long long)j * M + m], v, acc);
        }
        if (p.y_chan_major) p.y[((long long)b * M + m) * p.n + i] = acc;
        else p.y[((long long)b * p.n + i) * M + m] = acc;
        acc_mean += acc;
        if (p.diff && m > 0) p.diff[((long long)b * p.n + i) * (M - 1) + m - 1] = prev - acc;
        prev = acc;
    }
    if (p.mean) p.mean[(long long)b * p.n + i] = acc_mean / (float)M;
}

// new history = last L-1 samples of [cache ; x]
DS_HD void td_fir_cache(const TdParams& p, int b, int i) {
    const int M = p.M, L = p.L;
    const int s = i + p.n - (L - 1);                                    // position in x of history slot i (may be negative)
    for (int m = 0; m < M; ++m) {
        const float v = s >= 0 ? (p.x_chan_major ? p.x[((long long)b * M + m) * p.n + s] : p.x[((long long)b * p.n + s) * M + m])
                               : p.cache_in[((long long)b * M + m) * (L - 1) + (i + p.n)];
        p.cache_out[((long long)b * M + m) * (L - 1) + i] = v;
    }
}

}  // namespace ds
