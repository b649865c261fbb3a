// ds_pipe.hpp — the fused frame program of ds_core.hpp (Engine) as a hop-level software pipeline, for 512-point frames.
//
// Engine walks a hop as a chain of phases — forward transforms, split, per-bin recursion, inverse transform, overlap-add — and every
// link is either a handful of LDS round trips or one long run of arithmetic.  With 10 s of signal per call (625 hops per launch, SURVEY
// section 8d's chunked regime) the state stays in registers and the launch is bound by that chain: the vector units are ~2/3 busy and the
// LDS ~1/2 (profiles/r03a/), and for a third of a hop only one or two of the four waves have anything to do (the inverse transform is one
// wave's work, the overlap-add two waves').  The per-bin recursion of hop s is the only part that depends on hop s - 1's; the forward
// transforms of hop s + 1 and the inverse transform / overlap-add of hops s - 1 / s - 2 do not depend on it at all.  So here they run
// TOGETHER, stage by stage, inside one wave-local phase ("A"):
//
//   A(s):  every wave:  forward stage i of hop s + 1   |  part i of the per-bin program of hop s        i = 1 .. 4
//          wave 2 also: inverse stage i of hop s - 1;   waves 1, 3 also: overlap-add of hop s - 2;   wave 0 also: the Nyquist bin of hop s
//   barrier;  C(s): split of hop s + 1's spectra into the bins' registers (needs every channel);  barrier
//
// Every stage loads its operands first, the per-bin arithmetic runs while they are in flight, then the butterflies: the LDS latency of
// the transforms hides behind the arithmetic of the recursion in the same wave.  Two workgroup barriers per hop instead of five (plus
// eight wave-local hand-offs).  The transforms run IN PLACE (a channel's 64 butterflies per stage are one wave's, so between a stage's
// loads and its stores the whole channel sits in that wave's registers, and the LDS executes a wave's accesses in order): one forward
// buffer instead of two, which pays for the second inverse buffer and the second output spectrum — 35 KB of LDS, still four workgroups
// per CU.  The arithmetic is Engine's, operation for operation: the two give the same samples and the same state bit for bit
// (tests/test_kernel_emul.py::test_emul_pipelined_engine_equals_the_frame_engine, tests/test_gpu_parity.py::test_pipelined_kernel_*).
//
// Reference semantics: as ds_core.hpp (transform/transform.py:407-481, beamformer/adaptivebeamformer.py:44-128, beamformer/GSC.py:174-294,
// noise_estimation/mcra.py:27-77, noise_estimation/mc_mcra.py:91-224).
#pragma once
#include "ds_core.hpp"

namespace ds {

#if defined(__HIP_DEVICE_COMPILE__)
#define DS_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define DS_UNIFORM(x) (x)
#endif

template <int V> struct IntC { static constexpr int value = V; };
// the engine's helper lambdas are called from several places: they must be inlined wherever they are called, or the thread's register block
// (passed by reference) would have to live in memory
#define DS_INL __attribute__((always_inline))

template <int NFFT, int M, int NYQF> struct SharedPipe {
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2;
    static constexpr int NCP = NC + NC / 4;
    alignas(16) float xbuf[M][N];     // two halves: [old hop | new hop], roles swap every frame
    cf fa[M][NCP];                    // forward transforms of the hop in flight, in place
    cf fi[2][NCP];                    // inverse transform of hop s - 1 (in place) beside the overlap-add of hop s - 2
    alignas(16) Tables<NFFT> tb;
    float pw[K + 3];                  // |Z_0|^2 for the MCRA frequency stencil
    cf Y[2][K + 1];                   // beamformer output spectrum of hop s (written) and of hop s - 1 (being transformed back)
    float ynq[4];                     // the Nyquist bin's output of hops s .. s - 2 (ring)
    alignas(16) float tail[HOP];      // overlap-add tail
    alignas(16) float nyq[NYQF];      // per-bin state of the Nyquist bin (k = N/2)
    float zn[M];                      // ... and its input Z[N/2][m] (real)
};

// The per-bin program in four parts (ADAPTIVE with method MVDR: MCRA + covariance recursion + start of the sweep | its columns | result);
// everything else runs whole in part 0.  Same operations in the same order as Engine::bin_program.
template <int M, int ALGO, bool RYY> struct BinParts {
    typedef StateLayout<M, ALGO, RYY> SL;
    MvdrSweep<M> sw;
    cf acc;
    static constexpr int col_begin(int part) { return part == 1 ? 0 : part == 2 ? (M + 3) / 4 : part == 3 ? (3 * M + 3) / 4 : M; }

    template <int J, int JE> DS_HD void columns() {          // columns J .. JE - 1 of the sweep, each with its index a constant
        if constexpr (J < JE) { sw.column(J); columns<J + 1, JE>(); }
    }

    template <int NFFT> static DS_HD cf whole(float* st, const cf* Z, const cf* steer, int k, const float* pw, const Params& p, int frm_cnt,
                                              bool reset, int spp_cnt) {
        constexpr int K = NFFT / 2 + 1;
        cf a[M];
#pragma unroll
        for (int m = 0; m < M; ++m) a[m] = steer[k * M + m];
        if constexpr (ALGO == ALGO_FIXED) {
            return fixed_bin<M>(Z, a);
        } else if constexpr (ALGO == ALGO_ADAPTIVE) {
            mcra_bin(st + SL::MC_S, k, K, pw[k > 0 ? k - 1 : 0], pw[k], pw[k + 1], frm_cnt, reset, p.mcra_L);
            return adaptive_bin<M, RYY>(st, Z, a, p, nullptr, k);
        } else {
            return gsc_bin<M>(st, Z, a, p, k, spp_cnt);
        }
    }

    // MV: the method is MVDR (the call's method is hoisted out of the hop loop: a sweep that is only sometimes re-initialised would be live
    // around the loop, 37 registers for nothing)
    template <int NFFT, int PART, bool MV> DS_HD void part(float* st, const cf* Z, const cf* steer, int k, const float* pw, const Params& p,
                                                           int frm_cnt, bool reset, int spp_cnt) {
        constexpr int K = NFFT / 2 + 1;
        if constexpr (ALGO != ALGO_ADAPTIVE || !MV) {
            if constexpr (PART == 0) acc = whole<NFFT>(st, Z, steer, k, pw, p, frm_cnt, reset, spp_cnt);
        } else if constexpr (PART == 0) {
            cf a[M];
#pragma unroll
            for (int m = 0; m < M; ++m) a[m] = steer[k * M + m];
            mcra_bin(st + SL::MC_S, k, K, pw[k > 0 ? k - 1 : 0], pw[k], pw[k + 1], frm_cnt, reset, p.mcra_L);
            float* d = st + SL::R_DIAG;
            float* o = st + SL::R_OFF;
            if (RYY) herm_rank1<M>(st + SL::RYY_DIAG, st + SL::RYY_OFF, Z, p.alpha_y, p.beta_y);      // adaptive_bin, word for word
            if (st[SL::MC_S + 3] < p.gate && gate_open(p, k)) herm_rank1<M>(d, o, Z, p.alpha_v, p.beta_v);
            sw.init(d, o, p.diag, a, Z);
        } else {
            columns<col_begin(PART), col_begin(PART + 1)>();
            if constexpr (PART == 3) acc = sw.finish();
        }
    }
};

template <int M, int ALGO, bool RYY, int NPRE, int NJ> struct RegsPipe {
    cf Z[M];
    float st[StateLayout<M, ALGO, RYY>::NP * 4 + 1];
    vec4 pre[NPRE];
    const vec4* xp[NPRE];
    vec4 nyq;
    cf fv[NJ][4];             // forward butterfly operands of this thread's jobs, between a stage's loads and its stores
    vec4 fw[NJ];              // ... and the stage's twiddle pair
    cf iv[4];                 // the same for the inverse transform (wave INV_WAVE)
    vec4 iw;
    BinParts<M, ALGO, RYY> bp;
};

template <int NFFT, int M, int ALGO, bool RYY> struct PipeEngine {
    static_assert(NFFT == 512, "a channel's radix-4 stages must be one wavefront's work (64 butterflies per stage)");
    static_assert(ALGO == ALGO_FIXED || ALGO == ALGO_ADAPTIVE || ALGO == ALGO_GSC, "frame algorithms");
    typedef Engine<NFFT, M, ALGO, RYY> EB;              // shares the staging helpers and the state layout
    static constexpr int N = NFFT, NC = NFFT / 2, K = NFFT / 2 + 1, HOP = NFFT / 2;
    static constexpr int NT = NC, NB = NC / 4;               // 64 butterflies per channel and stage
    static constexpr int KP = EB::KP, NP = EB::NP, NV4 = EB::NV4, NPRE = EB::NPRE;
    static constexpr int NJ = (M * NB + NT - 1) / NT;        // forward jobs per thread and stage
    static constexpr int NYQ_TID = 0;                        // wave 0: the Nyquist bin
    static constexpr int INV_WAVE = 2;                       // wave 2: the inverse transform
    typedef typename EB::SL SL;
    typedef SharedPipe<NFFT, M, (SL::NP > 0 ? SL::NP * 4 : 4)> Sh;
    typedef RegsPipe<M, ALGO, RYY, NPRE, NJ> Rg;

    // the overlap-add runs on waves 1 and 3: sample pair i of the hop
    static DS_HD bool ola_lane(int tid, int& i) { i = (tid & 63) + ((tid >> 7) << 6); return ((tid >> 6) & 1) != 0; }

    template <int STAGE> static DS_HD void fwd_load(int tid, Sh& sh, int old_half, Rg& r) {
        constexpr int Ns = STAGE == 1 ? 1 : STAGE == 2 ? 4 : STAGE == 3 ? 16 : 64;
        constexpr int PIN = STAGE == 2 ? 1 : STAGE == 3 ? 2 : 0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int idx = tid + i * NT;
            if (idx < M * NB) {
                const int ch = (int)((unsigned)idx / (unsigned)NB), j = (int)((unsigned)idx % (unsigned)NB);
                fft_load<NFFT, 4, -1, (STAGE == 1 ? 1 : 0), PIN>(sh, &sh.fa[0][0], ch, j, Ns, old_half, r.fv[i], r.fw[i]);
            }
        }
    }
    template <int STAGE> static DS_HD void fwd_finish(int tid, Sh& sh, Rg& r) {
        constexpr int Ns = STAGE == 1 ? 1 : STAGE == 2 ? 4 : STAGE == 3 ? 16 : 64;
        constexpr int POUT = STAGE == 1 ? 1 : STAGE == 2 ? 2 : 0;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int idx = tid + i * NT;
            if (idx < M * NB) {
                const int ch = (int)((unsigned)idx / (unsigned)NB), j = (int)((unsigned)idx % (unsigned)NB);
                fft_finish<NFFT, 4, -1, POUT>(sh, &sh.fa[0][0], ch, j, Ns, r.fv[i], r.fw[i]);
            }
        }
    }
    // inverse stage of the hop whose spectrum is Y[buf], transform in fi[buf]; lanes of wave INV_WAVE only
    template <int STAGE> static DS_HD void inv_load(int j, Sh& sh, int buf, Rg& r) {
        constexpr int Ns = STAGE == 1 ? 1 : STAGE == 2 ? 4 : STAGE == 3 ? 16 : 64;
        constexpr int PIN = STAGE == 2 ? 1 : STAGE == 3 ? 2 : 0;
        if constexpr (STAGE == 1) fft_load<NFFT, 4, +1, 2, 0>(sh, &sh.Y[buf][0], 0, j, Ns, 0, r.iv, r.iw);
        else fft_load<NFFT, 4, +1, 0, PIN>(sh, &sh.fi[buf][0], 0, j, Ns, 0, r.iv, r.iw);
    }
    template <int STAGE> static DS_HD void inv_finish(int j, Sh& sh, int buf, Rg& r) {
        constexpr int Ns = STAGE == 1 ? 1 : STAGE == 2 ? 4 : STAGE == 3 ? 16 : 64;
        constexpr int POUT = STAGE == 1 ? 1 : STAGE == 2 ? 2 : 0;
        fft_finish<NFFT, 4, +1, POUT>(sh, &sh.fi[buf][0], 0, j, Ns, r.iv, r.iw);
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
        int frm_cnt = cnt[0], ell = cnt[1], spp_cnt = cnt[2];
        int old_half = 0;
        const int T = p.T;
        const bool wave_in = p.x_sample_stride == 1;          // [M][L] input: a channel's samples are staged by the wave that transforms it

        // ---- prologue: tables, tails, per-bin state (Engine's, word for word) -------------------------
        ex.phase([&](int tid, Rg& r) DS_INL {
            {
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
            prefetch_init(p, xb, tid, r);
            prefetch(p, tid, r);
#pragma unroll
            for (int q = 0; q < SL::NPF; ++q) {
                const vec4 v = load_state(&bins[q * KP + tid]);
                r.st[4 * q] = v.x; r.st[4 * q + 1] = v.y; r.st[4 * q + 2] = v.z; r.st[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int j = 0; j < SL::RT; ++j) r.st[4 * SL::NPF + j] = btail[tid * SL::RT + j];
            if constexpr (NP > 0) {                                     // Nyquist planes: parked in a register until the first split phase
                r.nyq = bins[(tid < SL::NPF ? tid : 0) * KP + NC];
                if (SL::RT > 0 && tid == SL::NPF) {
                    float w[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < SL::RT; ++j) w[j] = btail[NC * SL::RT + j];
                    r.nyq.x = w[0]; r.nyq.y = w[1]; r.nyq.z = w[2]; r.nyq.w = w[3];
                }
            }
        });

        auto ph = [&](bool wave_local, auto f) DS_INL { if (wave_local) ex.phase_wave(f); else ex.phase(f); };
        // hop t's samples into the new half of xbuf, hop t + 1's on their way into registers
        auto stage_in = [&](int t) DS_INL {
            const int new_half = old_half ^ 1;
            ph(wave_in, [&](int tid, Rg& r) DS_INL {
                commit(p, sh, new_half, tid, r);
                if (t + 1 < T) prefetch(p, tid, r);
            });
        };
        // split of the packed spectra of the hop whose transforms have just finished -> Z[k][m] in registers, |Z_0|^2 and the Nyquist bin's inputs
        auto split = [&](bool first) DS_INL {
            ex.phase([&](int tid, Rg& r) DS_INL {
                if (first && tid < NP) {                                // Nyquist planes -> LDS (loaded in the prologue)
                    sh.nyq[4 * tid] = r.nyq.x; sh.nyq[4 * tid + 1] = r.nyq.y; sh.nyq[4 * tid + 2] = r.nyq.z; sh.nyq[4 * tid + 3] = r.nyq.w;
                }
                const cf* F = &sh.fa[0][0];
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
                    for (int m = 0; m < M; ++m) {
                        r.Z[m].y = 0.0f;
                        const cf F0 = F[m * Sh::NCP];
                        sh.zn[m] = F0.x - F0.y;                         // the Nyquist bin's inputs
                    }
                    const float zn = sh.zn[0];
                    sh.pw[NC] = zn * zn;
                }
                sh.pw[k] = cabs2(r.Z[0]);
            });
        };

        if (T > 0) {
            // ---- hop 0: staged, transformed and split with nothing beside it ------------------------------
            stage_in(0);
            ex.phase_wave2([&](int tid, Rg& r) DS_INL { fwd_load<1>(tid, sh, old_half, r); }, [&](int tid, Rg& r) DS_INL { fwd_finish<1>(tid, sh, r); });
            ex.phase_wave2([&](int tid, Rg& r) DS_INL { fwd_load<2>(tid, sh, 0, r); }, [&](int tid, Rg& r) DS_INL { fwd_finish<2>(tid, sh, r); });
            ex.phase_wave2([&](int tid, Rg& r) DS_INL { fwd_load<3>(tid, sh, 0, r); }, [&](int tid, Rg& r) DS_INL { fwd_finish<3>(tid, sh, r); });
            ex.phase2([&](int tid, Rg& r) DS_INL { fwd_load<4>(tid, sh, 0, r); }, [&](int tid, Rg& r) DS_INL { fwd_finish<4>(tid, sh, r); });
            old_half ^= 1;
            split(true);
        }

        // ---- the pipeline: super-step s = bins of hop s | transforms of hop s + 1 | inverse transform of hop s - 1 | overlap-add of s - 2
        auto pipeline = [&](auto mv_tag) DS_INL {
        constexpr bool MV = decltype(mv_tag)::value != 0;
        for (int s = 0; s < T + 2 && T > 0; ++s) {
            const bool do_bins = s < T, do_fwd = s + 1 < T, do_inv = s >= 1 && s <= T, do_ola = s >= 2;
            const int cur = s & 1, prv = cur ^ 1;                       // Y / fi buffers of hop s and of hop s - 1 (= hop s - 2's: cur)
            const bool reset = (frm_cnt != 0) && (ell % p.mcra_L == 0);
            if (do_fwd) stage_in(s + 1);
            const int oh = old_half;
            // one sub-phase: the stage's loads first (forward, inverse), part PART of the per-bin program while they are in flight, then
            // the butterflies and stores
            auto sub = [&](auto stage_tag, auto extra) DS_INL {
                constexpr int STAGE = decltype(stage_tag)::value;
                ex.phase_wave2(
                    [&](int tid, Rg& r) DS_INL {
                        if (do_fwd) fwd_load<STAGE>(tid, sh, oh, r);
                        if (do_inv && DS_UNIFORM(tid >> 6) == INV_WAVE) inv_load<STAGE>(tid & 63, sh, prv, r);
                    },
                    [&](int tid, Rg& r) DS_INL {
                        if (do_bins) r.bp.template part<NFFT, STAGE - 1, MV>(r.st, r.Z, steer, tid, sh.pw, p, frm_cnt, reset, spp_cnt);
                        if (do_fwd) fwd_finish<STAGE>(tid, sh, r);
                        if (do_inv && DS_UNIFORM(tid >> 6) == INV_WAVE) inv_finish<STAGE>(tid & 63, sh, prv, r);
                        extra(tid, r);
                    });
            };
            auto nothing = [](int, Rg&) DS_INL {};
            sub(IntC<1>{}, [&](int tid, Rg&) DS_INL {
                int i;
                if (do_ola && ola_lane(tid, i)) {                       // window, overlap-add, emit hop s - 2
                    const cf* Zi = &sh.fi[cur][0];
                    const float sc = 1.0f / (float)NC;
                    const float hn = 0.5f * sh.ynq[(s - 2) & 3];       // the Nyquist bin's share of every even (+) / odd (-) sample
                    cf z1 = Zi[i], z2 = Zi[i + NC / 2];
                    z1.x += hn; z1.y -= hn; z2.x += hn; z2.y -= hn;
                    const float y0 = sh.tb.win[2 * i] * (z1.x * sc), y1 = sh.tb.win[2 * i + 1] * (z1.y * sc);
                    const float o0 = (y0 + sh.tail[2 * i]) * p.out_scale, o1 = (y1 + sh.tail[2 * i + 1]) * p.out_scale;
                    sh.tail[2 * i] = sh.tb.win[HOP + 2 * i] * (z2.x * sc);
                    sh.tail[2 * i + 1] = sh.tb.win[HOP + 2 * i + 1] * (z2.y * sc);
                    float* dst = p.y + yb + (long long)(s - 2) * HOP + 2 * i;
#if defined(__HIP_DEVICE_COMPILE__)
                    typedef float f2_t __attribute__((ext_vector_type(2)));
                    f2_t o; o.x = o0; o.y = o1;
                    __builtin_nontemporal_store(o, reinterpret_cast<f2_t*>(dst));
#else
                    dst[0] = o0; dst[1] = o1;
#endif
                }
            });
            sub(IntC<2>{}, nothing);
            sub(IntC<3>{}, nothing);
            sub(IntC<4>{}, [&](int tid, Rg& r) DS_INL {
                if (do_bins) {
                    cf Yk = r.bp.acc;
                    if (tid == 0) Yk.y = 0.0f;                          // irfft ignores Im Y[0] and Im Y[N/2]
                    sh.Y[cur][tid] = Yk;
                    if (tid == NYQ_TID) {                               // the Nyquist bin of hop s: state in LDS, one lane
                        cf Zn[M];
#pragma unroll
                        for (int m = 0; m < M; ++m) Zn[m] = mk(sh.zn[m], 0.0f);
                        const cf Yn = BinParts<M, ALGO, RYY>::template whole<NFFT>(sh.nyq, Zn, steer, NC, sh.pw, p, frm_cnt, reset, spp_cnt);
                        sh.ynq[s & 3] = Yn.x;                           // irfft ignores Im Y[N/2]
                    }
                }
            });
            if (do_fwd) old_half ^= 1;
            if (do_bins) {
                if (ALGO == ALGO_ADAPTIVE) {
                    if (reset) ell = 0;
                    frm_cnt += 1; ell += 1;
                }
                if (ALGO == ALGO_GSC) spp_cnt += 1;
            }
            ex.sync();                                                  // Y of hop s, the transforms of hop s + 1 and fi of hop s - 1 are complete
            if (do_fwd) split(false);                                   // (ends with a barrier: the next super-step's stores into fa / pw's readers)
        }
        };
        if (ALGO == ALGO_ADAPTIVE && p.method == METHOD_MVDR) pipeline(IntC<1>{}); else pipeline(IntC<0>{});

        // ---- epilogue: state back to HBM (Engine's, word for word) --------------------------------------
        ex.phase([&](int tid, Rg& r) DS_INL {
            {
                vec4* tin4 = reinterpret_cast<vec4*>(tin);
                for (int i = tid; i < M * HOP / 4; i += NT) {
                    const int m = i / (HOP / 4), q = i - m * (HOP / 4);
                    store_state(&tin4[i], *reinterpret_cast<const vec4*>(&sh.xbuf[m][old_half * HOP + 4 * q]));
                }
                vec4* tout4 = reinterpret_cast<vec4*>(tout);
                for (int i = tid; i < HOP / 4; i += NT) store_state(&tout4[i], *reinterpret_cast<const vec4*>(&sh.tail[4 * i]));
            }
#pragma unroll
            for (int q = 0; q < SL::NPF; ++q) {
                vec4 v; v.x = r.st[4 * q]; v.y = r.st[4 * q + 1]; v.z = r.st[4 * q + 2]; v.w = r.st[4 * q + 3];
                store_state(&bins[q * KP + tid], v);
            }
#pragma unroll
            for (int j = 0; j < SL::RT; ++j) btail[tid * SL::RT + j] = r.st[4 * SL::NPF + j];
            if (T > 0) {                                                // (a call without hops never moved the Nyquist planes into LDS)
                if (tid < SL::NPF) {
                    vec4 v; v.x = sh.nyq[4 * tid]; v.y = sh.nyq[4 * tid + 1]; v.z = sh.nyq[4 * tid + 2]; v.w = sh.nyq[4 * tid + 3];
                    bins[tid * KP + NC] = v;
                } else if (SL::RT > 0 && tid == SL::NPF) {
#pragma unroll
                    for (int j = 0; j < SL::RT; ++j) btail[NC * SL::RT + j] = sh.nyq[4 * SL::NPF + j];
                }
            }
            if (tid == 0) { cnt[0] = frm_cnt; cnt[1] = ell; cnt[2] = spp_cnt; }
        });
    }

    // staging helpers (Engine's, on this engine's register block)
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
    static DS_HD void prefetch(const Params& p, int tid, Rg& r) {
        const int step = p.x_sample_stride == 1 ? HOP / 4 : HOP * M / 4;   // vec4 per hop along this lane's stream
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            if (tid + i * NT < NV4) {
#if defined(__HIP_DEVICE_COMPILE__)
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
};

}  // namespace ds
