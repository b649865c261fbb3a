// ds_api_chains.hip — the chain handles: whole reference structures behind one native handle, every stage a kernel on the
// handle's stream reading the previous stage's device buffer.
//   DS_ALGO_WPE_MVDR     BASELINE config 4: STFT -> RLS-WPE -> McMcra gain -> adaptive MVDR frame loop -> ISTFT
//   DS_ALGO_SUBBAND_GSC  SubbandGSC.process (beamformer/SubbandGSC.py:170-262); BASELINE config 5 with RLS blocking filters
#include "ds_handle.hpp"

using namespace dsi;

namespace dsi {
// DS_ALGO_WPE_MVDR: STFT -> frame delay line -> WPE -> McMcra gain -> adaptive MVDR frame loop x gain -> ISTFT, every stage a
// kernel on h->stream reading the previous stage's device buffer (nothing returns to the host between the stages)
int chain_reserve(ds_handle* h, int T) {
    const size_t B = h->cfg.batch, K = h->K, M = h->cfg.n_mics, d = h->wpe_delay > 0 ? h->wpe_delay : 1;
    const size_t pg = h->wpe_only ? 0 : B * T * K * 4;              // Wpe.update alone: no speech-presence / gain arrays; [5] = channel 0 of E
    const size_t need[8] = {B * T * K * M * 8, 0, B * T * K * M * 8, pg, pg, B * T * K * 8, B * d * K * M * 8, 0};
    for (int i = 0; i < 8; ++i) {
        if (need[i] == 0 || need[i] <= h->chain_bytes[i]) continue;
        { const int jr = join_groups(h); if (jr) return jr; }
        DS_HIP(h, hipStreamSynchronize(h->stream));
        h->graph_valid = false; h->chain_warm_n = -1;
        (void)hipFree(h->chain_buf[i]); h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0;
        DS_HIP(h, hipMalloc((void**)&h->chain_buf[i], need[i]));
        h->chain_bytes[i] = need[i];
        if (i >= 6) DS_HIP(h, hipMemset(h->chain_buf[i], 0, need[i]));      // the stream starts from silence (DelaySamples, awpe.py:75-76)
    }
    return DS_OK;
}

#ifdef DS_ABLATE_CHAIN   // timing experiments only (scratch/jobs_r06/jobs_r06_abl.sh): the pipelined chain WITHOUT one of its stages' launches —
// what that stage costs the chain next to the others (its time alone says little: the stages share the chip).  Never in the shipped library
// DS_ABL_AFTER=<n>: the stage runs in the first n chain calls (pieces), so that what the others read afterwards is stale but REAL data —
// a stage that never ran leaves zeros behind, and the operators' data-dependent paths (Laguerre steps, fallbacks) then cost something else
static int g_abl_calls = 0;
static bool abl_skip(const char* stage) {
    const char* e = std::getenv("DS_ABL_SKIP");
    if (!e || std::strstr(e, stage) == nullptr) return false;
    const char* a = std::getenv("DS_ABL_AFTER");
    return g_abl_calls > (a ? std::atoi(a) : 0);
}
#define DS_ABL_RUN(stage) (!abl_skip(stage))
#else
#define DS_ABL_RUN(stage) true
#endif
// The same chain with the batch cut into `S` utterance groups, each running the whole chain on its own stream at its own pace: nothing
// joins the groups between calls (join_groups() does when anything else touches the handle), so one group's HBM-bound WPE kernel runs
// next to the other groups' analysis / McMcra / MVDR / synthesis stages, whichever step those are in.  Utterances never interact and
// every kernel gets its sub-range through its pointers; each group reads and advances its own copy of the device counters.
static int chain_process_groups(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                                int n_samples, float* y_dev, long long y_batch_stride, int S) {
    const int B = h->cfg.batch, M = h->cfg.n_mics, K = h->K, T = n_samples / h->cfg.hop, dl = h->wpe_delay;
    float *D = h->chain_buf[0], *E = h->chain_buf[2], *pp = h->chain_buf[3], *G = h->chain_buf[4], *Y = h->chain_buf[5];
#ifdef DS_ABLATE_CHAIN
    ++g_abl_calls;
#endif
    if (!h->ev_fork) DS_HIP(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int g = 0; g < S - 1; ++g) {
        if (!h->side[g]) DS_HIP(h, hipStreamCreateWithFlags(&h->side[g], hipStreamNonBlocking));
        if (!h->ev_join[g]) DS_HIP(h, hipEventCreateWithFlags(&h->ev_join[g], hipEventDisableTiming));
    }
    int rc = flush_tick(h); if (rc) return rc;
    if (!h->groups_open) {
        // first call since the groups were joined: whatever is on the chain's stream (state imports, input uploads, a replayed graph) comes
        // first for every group; from here on the groups only follow their own streams
        DS_HIP(h, hipEventRecord(h->ev_fork, h->stream));
        for (int g = 1; g < S; ++g) DS_HIP(h, hipStreamWaitEvent(h->side[g - 1], h->ev_fork, 0));
    }
    // (The groups of a call start together and their stages take the same time, so they run IN STEP: WPE next to WPE, operators next to
    // operators.  Round 6 held group 1 back by a WPE kernel once, so that a group's WPE ran next to the other's operators from then on: 8.78 M
    // against 8.83 M frames/s with 10 s per call — the stages' costs add up whichever way they overlap.  profiles/r06a/cfg4_stagger_ab.txt)
    const ds::TickArgs none = {nullptr, 0, 1, 0, 0};
    for (int g = 0; g < S; ++g) {
        const int lo = (int)((long long)B * g / S), nb = (int)((long long)B * (g + 1) / S) - lo;
        if (nb == 0) continue;
        hipStream_t sg = g == 0 ? h->stream : h->side[g - 1];
        const size_t oz = (size_t)lo * T * K * M * 2, o1 = (size_t)lo * T * K;
        {   // analysis of the group's utterances (io through the pointers, the carried tails through batch0)
            ds_handle* t = h->sub[0];
            Params p;
            fill_params(t, p);
            p.x = x_dev + (long long)lo * x_batch_stride; p.y = D + oz;
            p.x_batch_stride = x_batch_stride;
            p.y_batch_stride = (long long)T * K * M * 2;
            if (layout == DS_LAYOUT_CHANNELS_SAMPLES) { p.x_sample_stride = 1; p.x_chan_stride = x_chan_stride > 0 ? x_chan_stride : n_samples; }
            else { p.x_sample_stride = M; p.x_chan_stride = 1; }
            p.T = T; p.batch0 = lo;
            if (DS_ABL_RUN("stft")) DS_HIP(h, t->ki.launch(p, nb, sg));
        }
        if (!DS_ABL_RUN("wpe")) rc = 0;
        else if (dl > 0) rc = wpe_launch(h->sub[1], lo, nb, nullptr, D + oz, T, E + oz, h->chain_buf[6] + (size_t)lo * dl * K * M * 2, h->hist_cur, dl, h->dev_cnt + 8 * g + 3, sg);
        else rc = wpe_launch(h->sub[1], lo, nb, D + oz, D + oz, T, E + oz, nullptr, 0, 0, nullptr, sg);
        if (rc) return fail(h, rc, h->sub[1]->err);
        {
            const float* in[3] = {E + oz, nullptr, nullptr};
            float* out[5] = {pp + o1, G + o1, nullptr, nullptr, nullptr};
            rc = DS_ABL_RUN("mcmcra") ? binop_launch(h->sub[2], lo, nb, T, in, out, 0, 0, sg, none, g) : 0; if (rc) return fail(h, rc, h->sub[2]->err);
        }
        {
            const float* in[3] = {E + oz, G + o1, nullptr};
            float* out[5] = {Y + o1 * 2, nullptr, nullptr, nullptr, nullptr};
            rc = DS_ABL_RUN("mvdr") ? binop_launch(h->sub[3], lo, nb, T, in, out, 0, 1, sg, none, g) : 0; if (rc) return fail(h, rc, h->sub[3]->err);
        }
        {   // synthesis into the caller's rows; the group is addressed through the pointers so that the one-row-per-wavefront kernel applies,
            // and the launch carries the advance of the McMcra counters (the operator that read them is behind it on this stream)
            ds_handle* t = h->sub[4];
            Params p;
            fill_params(t, p);
            p.x = Y + o1 * 2; p.y = y_dev + (long long)lo * y_batch_stride;
            p.tail_out = t->tail_out + (size_t)lo * t->cfg.hop;
            p.x_batch_stride = (long long)T * K * 2;
            p.y_batch_stride = y_batch_stride;
            p.T = T; p.batch0 = 0; p.method = 1;
            p.tick = ds::TickArgs{h->sub[2]->dev_cnt + 8 * g, T, h->sub[2]->mcra_L > 0 ? h->sub[2]->mcra_L : 1, 0, 0};
            // ... and of the group's other counters, the WPE ring position and the frame loop's frame counter (their readers are behind this
            // launch on the stream as well): no counter kernel of its own at the end of a group's step (it was 6 us of each 390 us)
            p.tick2 = dl > 0 ? ds::TickArgs{h->dev_cnt + 8 * g, 0, 1, T % dl, dl} : none;
            p.tick3 = ds::TickArgs{h->sub[3]->dev_cnt + 8 * g, T, h->sub[3]->mcra_L > 0 ? h->sub[3]->mcra_L : 1, 0, 0};
            if (DS_ABL_RUN("istft")) DS_HIP(h, launch_transform_istft(t, p, nb, sg));
        }
    }
    h->groups_open = true;                                  // (a capture joins them before it ends: ds_process_device_seq)
    if (dl > 0) h->hist_cur = (h->hist_cur + T) % dl;
    advance_host_counters(h->sub[2], T, h->sub[2]->mcra_L);
    advance_host_counters(h->sub[3], T, h->sub[3]->mcra_L);
    return DS_OK;
}

int chain_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                                int n_samples, float* y_dev, long long y_batch_stride) {
    DS_HIP(h, hipSetDevice(h->device));                       // not set_device(): the utterance groups stay on their own streams between calls
    const int B = h->cfg.batch, M = h->cfg.n_mics, K = h->K, T = n_samples / h->cfg.hop;
    int rc = chain_reserve(h, T); if (rc) return rc;
    const int S = h->wpe_only ? 1 : (h->parts < B ? h->parts : B);
    if (S <= 1) { rc = join_groups(h); if (rc) return rc; }
    if (S > 1) return chain_process_groups(h, x_dev, layout, x_batch_stride, x_chan_stride, n_samples, y_dev, y_batch_stride, S);
    float *D = h->chain_buf[0], *E = h->chain_buf[2], *pp = h->chain_buf[3], *G = h->chain_buf[4], *Y = h->chain_buf[5];
#define DS_SUB(i, call) do { int rc_ = (call); if (rc_) return fail(h, rc_, h->sub[i]->err); } while (0)
    {   // analysis, strided input like the fused kernels take it
        ds_handle* t = h->sub[0];
        Params p;
        fill_params(t, p);
        p.x = x_dev; p.y = D;
        p.x_batch_stride = x_batch_stride;
        p.y_batch_stride = (long long)T * K * M * 2;
        if (layout == DS_LAYOUT_CHANNELS_SAMPLES) { p.x_sample_stride = 1; p.x_chan_stride = x_chan_stride > 0 ? x_chan_stride : n_samples; }
        else { p.x_sample_stride = M; p.x_chan_stride = 1; }
        p.T = T; p.batch0 = 0;
        DS_HIP(h, t->ki.launch(p, B, h->stream));
    }
    // delayed input of the prediction filter: a ring of the last wpe_delay analysis frames kept by the WPE kernel itself
    if (h->wpe_delay > 0) {
        // the ring position is device-resident (dev_cnt[3], advanced by the tick behind the launch) so that the step replays as a hipGraph
        DS_SUB(1, wpe_run(h->sub[1], nullptr, D, T, E, DS_MEM_DEVICE, h->chain_buf[6], h->hist_cur, h->wpe_delay, h->dev_cnt + 3, h->wpe_only ? Y : nullptr));
        rc = post_tick(h->sub[1], h->dev_cnt, 0, 1, T % h->wpe_delay, h->wpe_delay, h->stream); if (rc) return rc;   // rides in the McMcra launch
        h->hist_cur = (h->hist_cur + T) % h->wpe_delay;
    } else {
        DS_SUB(1, wpe_run(h->sub[1], D, D, T, E, DS_MEM_DEVICE, nullptr, 0, 0, nullptr, h->wpe_only ? Y : nullptr));
    }
    if (!h->wpe_only) {
        DS_SUB(2, ds_mcmcra_estimate(h->sub[2], E, T, pp, G, DS_MEM_DEVICE));
        DS_SUB(3, ds_adaptive_frames(h->sub[3], E, G, T, Y, DS_MEM_DEVICE));
    }
    {   // synthesis straight into the caller's (strided) output
        ds_handle* t = h->sub[4];
        Params p;
        fill_params(t, p);
        p.x = Y; p.y = y_dev;
        p.x_batch_stride = (long long)T * K * 2;
        p.y_batch_stride = y_batch_stride;
        p.T = T; p.batch0 = 0; p.method = 1;
        take_tick(t, h->stream, p.tick);                                // the adaptive frame loop's counter advance (Wpe.update alone: the delay line's)
        DS_HIP(h, launch_transform_istft(t, p, B, h->stream));
    }
#undef DS_SUB
    return flush_tick(h);
}

// ---- DS_ALGO_MCSPP_MVDR: the notebook's online MVDR (example/mvdr.ipynb cell 4) as a device-resident chain --------------------------
// chain_buf: [0] D complex [B][T][K][M] (transform.stft), [3] p [B][T][K] (noise_estimator.p), [5] Y complex [B][T][K] (Yout)
int nbmvdr_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride, int n_samples,
                          float* y_dev, long long y_batch_stride, float* p_dev) {
    int rc = set_device(h); if (rc) return rc;
#ifdef DS_ABLATE_CHAIN
    ++g_abl_calls;
#endif
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, K = h->K;
    const int T = n_samples / h->cfg.hop;
    if (h->sub[1]->aux_floats < K) return fail(h, DS_ESTATE, "McSpp-MVDR chain: call ds_chain_set_aux(DS_CHAIN_AUX_COHERENCE) first");
    // analysis with McCDR as its per-bin program (one launch less per call: the prior's kernel was 12 of a 120 us step at 6 microphones) where that
    // kernel exists; DS_CHAIN_UNFUSED=1 at ds_create keeps the separate launch (A/B runs and the reference point of the equality test)
    const bool cdr_fused = h->ki_cdr.launch != nullptr;
    const size_t need[8] = {B * T * K * M * 8, cdr_fused ? (B * T * K + B * T) * 4 : 0, 0, p_dev ? 0 : B * T * K * 4, 0, B * T * K * 8, 0, 0};
    for (int i = 0; i < 8; ++i) {
        if (need[i] == 0 || need[i] <= h->chain_bytes[i]) continue;
        DS_HIP(h, hipStreamSynchronize(h->stream));
        h->graph_valid = false; h->chain_warm_n = -1;
        (void)hipFree(h->chain_buf[i]); h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0;
        DS_HIP(h, hipMalloc((void**)&h->chain_buf[i], need[i]));
        h->chain_bytes[i] = need[i];
    }
    float *D = h->chain_buf[0], *P = p_dev ? p_dev : h->chain_buf[3], *Y = h->chain_buf[5];
    // (round 6: the batch as two or three utterance groups on their own streams, as the frame kernels and the WPE chain have them, was built —
    // bit-identical — and measured: - 8 % with two groups, worse with three (profiles/r06a/nb_mvdr_utterance_groups_ab.txt): the operator's
    // launch already fills every wave slot twice over and the groups only add launches.  Removed.)
    // (also round 6: the three kernels of a call on three streams with two sets of the buffers between them, so that call t + 1's analysis +
    // McCDR and call t - 1's synthesis run next to call t's operator — which alone is 64 of the 91 us of a 6-microphone step, 28 of 46 at 4
    // (profiles/r06a/nb_mvdr_stage_ablation.txt).  Built, bit-identical over 36 one-block calls, resets and checkpoints, and measured:
    // one block per call 94 against 90 us at 6 microphones, 52 - 56 against 46 us at 4 — the four event waits per call cost more than the
    // overlap returns, since the operator already fills the chip; + 5 % at 6 microphones from 4 blocks per call on, nothing at 4.  Removed.
    // profiles/r06a/nb_mvdr_pipeline_ab.txt)
    {   // D = transform.stft(x)
        ds_handle* t = h->sub[0];
        Params p;
        fill_params(t, p);
        p.x = x_dev; p.y = D;
        p.x_batch_stride = x_batch_stride;
        p.y_batch_stride = (long long)T * K * M * 2;
        if (layout == DS_LAYOUT_CHANNELS_SAMPLES) { p.x_sample_stride = 1; p.x_chan_stride = x_chan_stride > 0 ? x_chan_stride : n_samples; }
        else { p.x_sample_stride = (int)M; p.x_chan_stride = 1; }
        p.T = T; p.batch0 = 0;
        if (cdr_fused) {                                                 // McCDR.estimation (mccdr.py:122-177) on the frame the workgroup has just analysed
            ds_handle* sp = h->sub[1];
            p.cdr_st = sp->opst; p.cdr_NF = sp->NF; p.cdr_frm = sp->op_frm; p.cdr_ell = sp->op_ell; p.cdr_L = 65; p.cdr_fn = sp->dev_buf[9];
            p.cdr_cnt = sp->use_dev_cnt ? sp->dev_cnt : nullptr;         // the counters as the device holds them: the sequence replays as a hipGraph
            p.cdr_gamma = h->chain_buf[1]; p.cdr_qavg = h->chain_buf[1] + B * T * K;
        }
        take_tick(t, h->stream, p.tick);
        if (DS_ABL_RUN("nbcdr")) DS_HIP(h, (cdr_fused ? h->ki_cdr.launch : t->ki.launch)(p, (int)B, h->stream));
    }
    // per frame: noise_estimator.estimation(y); steer = steering(Phi_xx); w = compute_mvdr_weight(steer, Phi_vv_inv); Yout = w^H y — one kernel
    if (cdr_fused) rc = mcspp_from_gamma(h->sub[1], D, T, h->chain_buf[1], h->chain_buf[1] + B * T * K, P, nullptr, nullptr, nullptr, Y);
    else rc = ds_mcspp_estimate(h->sub[1], D, T, P, nullptr, Y, nullptr, nullptr, DS_MEM_DEVICE);
    if (rc) return fail(h, rc, h->sub[1]->err);
    {   // yout = transform.istft(Yout)
        ds_handle* t = h->sub[2];
        Params p;
        fill_params(t, p);
        p.x = Y; p.y = y_dev;
        p.x_batch_stride = (long long)T * K * 2;
        p.y_batch_stride = y_batch_stride;
        p.T = T; p.batch0 = 0; p.method = 1;
        take_tick(t, h->stream, p.tick);                                // McSpp's counter advance
        if (DS_ABL_RUN("nbistft")) DS_HIP(h, launch_transform_istft(t, p, (int)B, h->stream));
        else { const int rc2 = flush_tick(h); if (rc2) return rc2; }
    }
    return flush_tick(h);
}

// ---- DS_ALGO_SUBBAND_GSC: SubbandGSC.process (SubbandGSC.py:170-262) as a device-resident chain ------------------------------
// (the buffer indices G_* are declared in ds_handle.hpp)
int chain2_reserve(ds_handle* h, int n) {
    const size_t B = h->cfg.batch, K = h->K, M = h->cfg.n_mics, T = n / h->cfg.hop, hop = h->cfg.hop;
    size_t need[G_COUNT] = {0};
    need[G_XN] = need[G_XA] = need[G_BM] = B * M * n * 4; need[G_FIXED] = B * n * 4;
    need[G_D] = need[G_XAIC] = B * T * K * M * 8; need[G_P] = B * T * K * 4;
    need[G_F] = need[G_E2] = B * T * K * 8; need[G_E] = B * M * T * K * 8;
    need[G_FPREV] = B * K * 8; need[G_FIXPREV] = B * hop * 4;
    if (h->ki_aic.launch) need[G_XAIC] = need[G_E2] = need[G_BM] = 0;         // the fused tail keeps all three in registers / LDS
    // second set of the front end's buffers: block t + 1's front end next to block t's later stages (not for very long calls, where the
    // front end is a small share of a call anyway and the set would cost gigabytes)
    const size_t second = 2 * need[G_XN] + need[G_FIXED] + need[G_D];
    if (h->ki_cdr.launch) need[G_GAM] = B * T * K * 4 + B * T * 4;
    if (h->front_async && second <= ((size_t)4 << 30)) {
        need[G_XN2] = need[G_XA2] = need[G_XN]; need[G_FIXED2] = need[G_FIXED]; need[G_D2] = need[G_D]; need[G_GAM2] = need[G_GAM];
        if (h->ki_aic.launch && h->tail_async) { need[G_P2] = need[G_P]; need[G_F2] = need[G_F]; need[G_EB2] = need[G_E]; }
    }
    for (int i = 0; i < G_COUNT; ++i) {
        if (need[i] == 0 || need[i] <= h->chain_bytes[i]) continue;
        { const int jr = join_groups(h); if (jr) return jr; }
        DS_HIP(h, hipStreamSynchronize(h->stream));
        h->graph_valid = false; h->chain_warm_n = -1;
        (void)hipFree(h->chain_buf[i]); h->chain_buf[i] = nullptr; h->chain_bytes[i] = 0;
        DS_HIP(h, hipMalloc((void**)&h->chain_buf[i], need[i]));
        h->chain_bytes[i] = need[i];
        if (i >= G_FPREV) DS_HIP(h, hipMemset(h->chain_buf[i], 0, need[i]));       // delay_fbf starts from silence (SubbandGSC.py:111)
    }
    return DS_OK;
}

// launch the STFT of sub-handle `t` on dense channel-major input x [batch][C][n] -> Y [batch][T][K][C]
static int chain_stft(ds_handle* h, ds_handle* t, const float* x, int n, float* Y) {
    Params p;
    fill_params(t, p);
    const int C = t->cfg.n_mics, T = n / t->cfg.hop;
    p.x = x; p.y = Y;
    p.x_batch_stride = (long long)C * n; p.x_sample_stride = 1; p.x_chan_stride = n;
    p.y_batch_stride = (long long)T * t->K * C * 2;
    p.T = T; p.batch0 = 0;
    take_tick(t, t->stream, p.tick);
    DS_HIP(h, launch_transform_stft(t, p, t->cfg.batch, t->stream));
    return DS_OK;
}
static int chain_istft(ds_handle* h, ds_handle* t, const float* Y, int T, float* y, long long y_batch_stride) {
    Params p;
    fill_params(t, p);
    p.x = Y; p.y = y;
    p.x_batch_stride = (long long)T * t->K * 2;
    p.y_batch_stride = y_batch_stride;
    p.T = T; p.batch0 = 0; p.method = 1;
    take_tick(t, t->stream, p.tick);
    DS_HIP(h, launch_transform_istft(t, p, t->cfg.batch, t->stream));
    return DS_OK;
}

// x_dev: [B][M][n] with element strides (x_bstride, x_cstride); y_dev [B] rows of n with stride y_bstride; the optional outputs dense
static int chain2_block(ds_handle* h, const float* x_dev, long long x_bstride, long long x_cstride, int n, float* y_dev, long long y_bstride,
                        float* fix_dev, float* bm_dev, float* p_dev, float* al_dev);

// A long call (BASELINE config 5's 10 s chunk = 625 blocks) runs as pieces of at most DS_CHAIN2_PIECE blocks: the stages' intermediate
// spectra are sized by a piece, not by the call (1.6 GB instead of 16 GB at 2048 utterances), the second buffer set exists, and the
// stage pipeline runs across the pieces — 62 blocks per piece was the best of {4 ... 125} (24.3 M frames/s against 22.1 M as one piece).
// A call of T blocks is T one-block calls by definition, so the pieces change nothing in the samples or the state.
constexpr int DS_CHAIN2_PIECE = 62;
int chain2_run(ds_handle* h, const float* x_dev, long long x_bstride, long long x_cstride, int n, float* y_dev, long long y_bstride,
               float* fix_dev, float* bm_dev, float* p_dev, float* al_dev) {
    const int hop = h->cfg.hop, T = n / hop;
    // (the optional outputs are dense arrays whose rows are as long as the call: a call that asks for them runs as one piece)
    if (T <= DS_CHAIN2_PIECE + DS_CHAIN2_PIECE / 2 || !h->front_async || fix_dev || bm_dev || p_dev || al_dev)
        return chain2_block(h, x_dev, x_bstride, x_cstride, n, y_dev, y_bstride, fix_dev, bm_dev, p_dev, al_dev);
#ifdef DS_ABLATE_CHAIN
    static const int piece_len = std::getenv("DS_ABL_PIECE") ? std::atoi(std::getenv("DS_ABL_PIECE")) : DS_CHAIN2_PIECE;
    const int pieces = (T + piece_len - 1) / piece_len;
#else
    const int pieces = (T + DS_CHAIN2_PIECE - 1) / DS_CHAIN2_PIECE;
#endif
    for (int i = 0, t0 = 0; i < pieces; ++i) {
        const int tn = T / pieces + (i < T % pieces ? 1 : 0);                 // the longer pieces first: the buffers are sized once
        const int rc = chain2_block(h, x_dev + (size_t)t0 * hop, x_bstride, x_cstride, tn * hop, y_dev + (size_t)t0 * hop, y_bstride,
                                    nullptr, nullptr, nullptr, nullptr);
        if (rc) return rc;
        t0 += tn;
    }
    return DS_OK;
}

static int chain2_block(ds_handle* h, const float* x_dev, long long x_bstride, long long x_cstride, int n, float* y_dev, long long y_bstride,
                        float* fix_dev, float* bm_dev, float* p_dev, float* al_dev) {
    DS_HIP(h, hipSetDevice(h->device));                      // not set_device(): the tail of the previous block stays on its own stream
#ifdef DS_ABLATE_CHAIN
    ++g_abl_calls;
#endif
    int rc = DS_OK;
    ds_handle* fe = h->sub[0];
    const int B = h->cfg.batch, M = h->cfg.n_mics, K = h->K, hop = h->cfg.hop, T = n / hop;
    if (fe->aux_floats == 0 || fe->aux_floats % M != 0 || h->sub[2]->aux_floats < (size_t)K)
        return fail(h, DS_ESTATE, "SubbandGSC chain: call ds_chain_set_aux(DS_CHAIN_AUX_FIR) and (DS_CHAIN_AUX_COHERENCE) first");
    rc = chain2_reserve(h, n); if (rc) return rc;
    float* cb[G_COUNT];
    for (int i = 0; i < G_COUNT; ++i) cb[i] = h->chain_buf[i];
    // the front end's stream and buffer set.  With a second set, block t + 1's front end only waits for the set to be free (block t - 1
    // behind it); otherwise it is ordered behind the whole of block t
    hipStream_t fs = h->front_async ? h->side[1] : h->stream;
    int set = 0;
    bool early = false;
    if (h->front_async) {
        const bool two = h->chain_buf[G_D2] != nullptr;
        if (!h->front_open) {                               // first block since something else used the chain's stream: start from there
            DS_HIP(h, hipEventRecord(h->ev_fork, h->stream));
            DS_HIP(h, hipStreamWaitEvent(fs, h->ev_fork, 0));
            h->front_open = true; h->fr_valid[0] = h->fr_valid[1] = false; h->tf_valid[0] = h->tf_valid[1] = false;
            h->bf_valid[0] = h->bf_valid[1] = false;
        }
        set = two ? h->front_set : 0;
        h->front_set ^= 1;
        // who read this set last: the analysis outputs (D, Gamma) are read by the middle stages, the fixed-beamformer block by the
        // blocking-filter branch, the notch / FIR outputs only by the front end itself (and by the copy of `aligned`, if the caller asked for
        // it).  When the branch joins at the tail (below) each front-end kernel waits for its own readers only, so that the notch and the
        // FIR bank of block t + 2 run next to the middle stages of block t and only the analysis waits for them
        early = h->early_front && two && h->lean_main && h->tail_async && h->ki_aic.launch && h->sub[5]->stream != h->stream && !h->al_read[set];
        if (!early) {
            if (h->fr_valid[set]) DS_HIP(h, hipStreamWaitEvent(fs, h->ev_fr[(h->fr_mid[set] ? 4 : 2) + set], 0));
            if (h->bf_valid[set]) DS_HIP(h, hipStreamWaitEvent(fs, h->ev_fr[8 + set], 0));     // ... and its blocking-filter branch (lean_main)
        }
        if (set) {
            cb[G_XN] = cb[G_XN2]; cb[G_XA] = cb[G_XA2]; cb[G_FIXED] = cb[G_FIXED2]; cb[G_D] = cb[G_D2]; cb[G_GAM] = cb[G_GAM2];
            if (h->chain_buf[G_P2]) { cb[G_P] = cb[G_P2]; cb[G_F] = cb[G_F2]; cb[G_E] = cb[G_EB2]; }
        }
    }
    const bool tail_async = h->tail_async && h->ki_aic.launch != nullptr;    // the tail on its own stream (side[2])
#define DS_SUB(i, call) do { int rc_ = (call); if (rc_) return fail(h, rc_, h->sub[i]->err); } while (0)
    const int Lt = (int)(fe->aux_floats / M);
    rc = frontend_set_taps(fe, Lt); if (rc) return fail(h, rc, fe->err);
    const bool cdr_in_front = h->ki_cdr.launch != nullptr;
    // the whole front end as ONE kernel (DC notch -> FIR bank + mean -> analysis + McCDR on the hop in LDS): where the shape can hold the
    // bank's history and windows in the analysis kernel's LDS, and nobody asks for the aligned channels as an array of their own
    const ds::KernelInfo kf = (h->front_fused && cdr_in_front && !al_dev) ? ds::lookup_front(h->cfg.nfft, M, Lt) : ds::KernelInfo{nullptr, 0, 0, 0};
    if (kf.launch) {
        if (early) {                                                    // one kernel: it waits for every reader of the set
            if (h->fr_valid[set]) DS_HIP(h, hipStreamWaitEvent(fs, h->ev_fr[(h->fr_mid[set] ? 4 : 2) + set], 0));
            if (h->bf_valid[set]) DS_HIP(h, hipStreamWaitEvent(fs, h->ev_fr[8 + set], 0));
        }
        ds_handle *t = h->sub[1], *sp = h->sub[2];
        Params p;
        fill_params(t, p);
        p.x = x_dev; p.y = cb[G_D];
        p.x_batch_stride = x_bstride ? x_bstride : (long long)M * n; p.x_sample_stride = 1; p.x_chan_stride = x_bstride ? x_cstride : n;
        p.y_batch_stride = (long long)T * K * M * 2;
        p.T = T; p.batch0 = 0;
        p.cdr_st = sp->opst; p.cdr_NF = sp->NF; p.cdr_frm = sp->op_frm; p.cdr_ell = sp->op_ell; p.cdr_L = 65; p.cdr_fn = sp->dev_buf[9];
        p.cdr_gamma = cb[G_GAM]; p.cdr_qavg = cb[G_GAM] + (size_t)B * T * K;
        // the FIR history's ping-pong halves by value (the host mirror of the parity: a pipelined chain is never replayed as a graph); the
        // device copy of the parity still flips, in this launch, so that the separate kernels could take over at any block
        p.fe_coef = fe->dev_buf[9]; p.fe_L = Lt; p.fe_mem = fe->td_mem; p.fe_radius = ds::decimal_double(fe->cfg.filt_alpha); p.fe_fixed = cb[G_FIXED];
        p.fe_cache_in = fe->td_cache[fe->td_cur]; p.fe_cache_out = fe->td_cache[fe->td_cur ^ 1];
        rc = post_tick(fe, fe->dev_cnt, 0, 1, 1, 2, fs); if (rc) return rc;
        fe->td_cur ^= 1;
        take_tick(t, fs, p.tick);
        DS_HIP(h, kf.launch(p, B, fs));
    } else {
    {   // :177-178 DC notch per channel, then :201,206 TimeAlignment FIR bank + channel mean (the fixed beamformer)
        ds::TdParams p;
        std::memset(&p, 0, sizeof p);
        p.B = B; p.M = M; p.n = n; p.x = x_dev; p.x_bstride = x_bstride; p.x_cstride = x_cstride; p.y = cb[G_XN]; p.mem = fe->td_mem;
        p.radius = ds::decimal_double(fe->cfg.filt_alpha);
        if (DS_ABL_RUN("notch")) DS_HIP(h, ds::launch_dcnotch(p, fs));
        if (early && h->bf_valid[set]) DS_HIP(h, hipStreamWaitEvent(fs, h->ev_fr[8 + set], 0));    // the FIR bank writes the fixed-beamformer block
        std::memset(&p, 0, sizeof p);
        p.B = B; p.M = M; p.n = n; p.L = Lt; p.x = cb[G_XN]; p.x_chan_major = 1; p.y = cb[G_XA]; p.y_chan_major = 1; p.mean = cb[G_FIXED];
        // the ping-pong parity of the FIR history is device-resident (dev_cnt[3], flipped by the tick behind the launch): graph replay
        p.coef = fe->dev_buf[9]; p.cache_in = fe->td_cache[0]; p.cache_out = fe->td_cache[1]; p.dev_parity = fe->dev_cnt + 3;
        if (DS_ABL_RUN("fir")) DS_HIP(h, ds::launch_fir(p, fs));
        rc = post_tick(fe, fe->dev_cnt, 0, 1, 1, 2, fs); if (rc) return rc;            // rides in the analysis launch that follows
        fe->td_cur ^= 1;
    }
    if (early && h->fr_valid[set]) DS_HIP(h, hipStreamWaitEvent(fs, h->ev_fr[(h->fr_mid[set] ? 4 : 2) + set], 0));        // the analysis writes D and Gamma
    if (cdr_in_front) {
        // :204 D, and McCDR (mcspp.py:250: the prior of McSpp) in the same kernel: thread k has bin k of microphones 0..2 in registers, the
        // MCRA stencil and the band mean of 1 - Gamma come from LDS; the counters go in by value (the host mirror of the McSpp stage)
        ds_handle *t = h->sub[1], *sp = h->sub[2];
        Params p;
        fill_params(t, p);
        p.x = cb[G_XA]; p.y = cb[G_D];
        p.x_batch_stride = (long long)M * n; p.x_sample_stride = 1; p.x_chan_stride = n;
        p.y_batch_stride = (long long)T * K * M * 2;
        p.T = T; p.batch0 = 0;
        p.cdr_st = sp->opst; p.cdr_NF = sp->NF; p.cdr_frm = sp->op_frm; p.cdr_ell = sp->op_ell; p.cdr_L = 65; p.cdr_fn = sp->dev_buf[9];
        p.cdr_gamma = cb[G_GAM]; p.cdr_qavg = cb[G_GAM] + (size_t)B * T * K;
        take_tick(t, fs, p.tick);
        if (DS_ABL_RUN("cdr")) DS_HIP(h, h->ki_cdr.launch(p, B, fs));
    } else {
        rc = chain_stft(h, h->sub[1], cb[G_XA], n, cb[G_D]); if (rc) return rc;                               // :204  D
    }
    }
    const bool fork = h->sub[5]->stream != h->stream;                       // blocking-filter branch on the side stream (RLS filters, see ds_create)
    const bool fused_tail = h->ki_aic.launch != nullptr;
    // with the tail on its own stream the blocking-filter branch joins THERE, and forks from the front end, not from the chain's stream:
    // every record / wait on the chain's stream is a barrier packet (~5 us) between two McSpp launches, which are the chain's critical
    // loop — that stream then carries one wait per dependency and one record per block, no join, no counter advance, no copy
    const bool join_at_tail = h->front_async && fork && fused_tail && tail_async && h->lean_main;
    if (h->front_async) {
        DS_HIP(h, hipEventRecord(h->ev_fr[set], fs));
        DS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_fr[set], 0));
        // the middle stages write p, F and the blocking-matrix outputs of this set: the tail that read them last comes first
        if (tail_async && h->tf_valid[set]) DS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_fr[6 + set], 0));
    }
    if (fork) {
        if (join_at_tail) {                                                  // the RLS filters do not take p: the branch follows the front end and the
            DS_HIP(h, hipStreamWaitEvent(h->sub[5]->stream, h->ev_fr[set], 0));                       // last tail of this set directly
            if (h->tf_valid[set]) DS_HIP(h, hipStreamWaitEvent(h->sub[5]->stream, h->ev_fr[6 + set], 0));
        } else {
            DS_HIP(h, hipEventRecord(h->ev_fork, h->stream));
            DS_HIP(h, hipStreamWaitEvent(h->sub[5]->stream, h->ev_fork, 0));
        }
    }
    // Round 5, SHELVED (make SHELVED=1 + DS_CHAIN_FAN_FUSED=1): the M RLS blocking filters INSIDE the McSpp launch (OP_MCSPP_STEADY_FAN).
    // They read the same frame of the same bin as McSpp does and do not take its p, so one thread can run both: the spectra D are fetched
    // once instead of twice and the filters' kernel — a streaming pass with 4 % of its cycles in arithmetic — disappears.  Built,
    // bit-identical (the chain-variant test), and measured: 9 % slower at one block per call (the pass had been running BESIDE McSpp on the
    // branch stream; fused it sits on the chain's critical stream), no change with 10 s per call — the hand-off bytes are not what bounds
    // that regime (profiles/r05a/cfg5_fan_in_mcspp_ab.txt)
    const bool fan_fused = h->fan_fused && cdr_in_front && fork && h->sub[5]->cfg.algo == DS_ALGO_SUBRLS && h->sub[5]->filter_len == 2 &&
                           h->sub[2]->op_frm >= 5 && !h->sub[2]->mcspp_repeat && (size_t)B * M * T * K * 8 < ((size_t)1 << 32);
    if (fan_fused) {
        rc = chain_stft(h, h->sub[3], cb[G_FIXED], n, cb[G_F]); if (rc) return rc;                               // bm[m].transform_x: F (branch stream)
        DS_HIP(h, hipEventRecord(h->ev_join[0], h->sub[3]->stream));
        DS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join[0], 0));
        DS_SUB(2, mcspp_from_gamma(h->sub[2], cb[G_D], T, cb[G_GAM], cb[G_GAM] + (size_t)B * T * K, cb[G_P], h->sub[5], cb[G_F], cb[G_E]));   // :208 p, :217-223 E
        advance_host_counters(h->sub[5], T, h->sub[5]->mcra_L);
    } else {
    if (!DS_ABL_RUN("mcspp")) advance_host_counters(h->sub[2], T, h->sub[2]->mcra_L);
    else if (cdr_in_front) DS_SUB(2, mcspp_from_gamma(h->sub[2], cb[G_D], T, cb[G_GAM], cb[G_GAM] + (size_t)B * T * K, cb[G_P]));   // :208  p
    else DS_SUB(2, ds_mcspp_estimate(h->sub[2], cb[G_D], T, cb[G_P], nullptr, nullptr, nullptr, nullptr, DS_MEM_DEVICE));
    if (DS_ABL_RUN("rows")) { rc = chain_stft(h, h->sub[3], cb[G_FIXED], n, cb[G_F]); if (rc) return rc; }                                   // bm[m].transform_x: F
    // :217-223 the M blocking filters: reference input F (shared), desired signal = channel m of D (bm[m].transform_d's analysis of the
    // aligned channel is the same spectrum), update probability p
    if (!DS_ABL_RUN("fan")) {}
    else if (h->sub[5]->cfg.algo == DS_ALGO_SUBRLS) DS_SUB(5, ds_subrls_update(h->sub[5], cb[G_F], cb[G_D], T, cb[G_E], DS_MEM_DEVICE));
    else DS_SUB(5, ds_sublms_update(h->sub[5], cb[G_F], cb[G_D], cb[G_P], T, cb[G_E], DS_MEM_DEVICE));
    }
    if (!fused_tail) {
        rc = chain_istft(h, h->sub[4], cb[G_E], T, cb[G_BM], n); if (rc) return rc;                           // bm outputs, [B*M][n] = [B][M][n]
        rc = chain_stft(h, h->sub[6], cb[G_BM], n, cb[G_XAIC]); if (rc) return rc;                              // :230-234  aic transform_x
    }
    hipStream_t cs = join_at_tail ? h->sub[5]->stream : h->stream;             // the stream of the carried-block copies below
    {   // fix_output = fixed beamformer output delayed by one block (:226,255); the carried block is state either way
        const size_t blk = (size_t)hop * 4, row = (size_t)n * 4;
        if (fork && !join_at_tail) {
            DS_HIP(h, hipEventRecord(h->ev_join[0], h->sub[5]->stream));
            DS_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join[0], 0));
        }
        if (fix_dev) {
            if (T > 1) DS_HIP(h, hipMemcpy2DAsync((char*)fix_dev + blk, row, cb[G_FIXED], row, row - blk, B, hipMemcpyDeviceToDevice, cs));
            DS_HIP(h, hipMemcpy2DAsync(fix_dev, row, cb[G_FIXPREV], blk, blk, B, hipMemcpyDeviceToDevice, cs));
        }
        DS_HIP(h, hipMemcpy2DAsync(cb[G_FIXPREV], blk, (char*)cb[G_FIXED] + (row - blk), row, blk, B, hipMemcpyDeviceToDevice, cs));
        if (join_at_tail) {
            DS_HIP(h, hipEventRecord(h->ev_fr[8 + set], h->sub[5]->stream));
            h->bf_valid[set] = true;
            h->groups_open = true;                                              // join_groups(): the branch comes back before anything else touches the handle
        }
    }
    // :226 delay_fbf: the canceller's desired signal is the fixed output one block late = F one frame late; the operator keeps the
    // carried frame in cb[G_FPREV] itself
    if (fused_tail) {
        // :224-262 as ONE frame kernel: synthesis of the M blocking-matrix outputs from the blocking filters' error spectra, their
        // re-analysis, the canceller as the per-bin program (state where the subband-LMS stage keeps it), synthesis into the caller's
        // rows — the blocking-matrix outputs (unless the caller asks for them), X_aic and the error spectrum never leave the workgroup
        ds_handle *ta = h->sub[6], *op = h->sub[7], *ts = h->sub[8];
        Params p;
        fill_params(ta, p);
        p.x = nullptr; p.y = y_dev;
        p.aic_e = cb[G_E]; p.aic_bmtail = h->sub[4]->tail_out; p.aic_bm = bm_dev;
        p.x_batch_stride = (long long)M * n; p.x_sample_stride = 1; p.x_chan_stride = n;
        p.y_batch_stride = y_bstride;
        p.T = T; p.batch0 = 0;
        p.tail_out = ts->tail_out; p.out_scale = ts->out_scale;
        p.aic_st = op->opst; p.aic_NF = op->NF; p.aic_d = cb[G_F]; p.aic_dprev = cb[G_FPREV]; p.aic_p = cb[G_P];
        p.aic_pc = op->p_complement; p.aic_norm = op->norm; p.aic_mu = op->filt_mu; p.aic_alpha = op->filt_alpha; p.aic_reg = 1e-4f;
        if (tail_async) {
            // McSpp's counter advance stays on the chain's stream (the next block's McSpp follows it there); the tail follows the middle stages
            // on its own stream, the next block's middle stages do not wait for it
            rc = flush_tick(h); if (rc) return rc;
            DS_HIP(h, hipEventRecord(h->ev_fr[4 + set], h->stream));
            DS_HIP(h, hipStreamWaitEvent(h->side[2], h->ev_fr[4 + set], 0));
            if (join_at_tail) DS_HIP(h, hipStreamWaitEvent(h->side[2], h->ev_fr[8 + set], 0));
            if (DS_ABL_RUN("tail")) DS_HIP(h, h->ki_aic.launch(p, B, h->side[2]));
            DS_HIP(h, hipEventRecord(h->ev_fr[6 + set], h->side[2]));
            h->tf_valid[set] = true;
            h->groups_open = true;                                          // join_groups(): side[2] comes back before anything else touches the handle
        } else {
            take_tick(op, h->stream, p.tick);                               // McSpp's counter advance
            DS_HIP(h, h->ki_aic.launch(p, B, h->stream));
        }
    } else {
        h->sub[7]->d_prev = cb[G_FPREV];
        DS_SUB(7, ds_sublms_update(h->sub[7], cb[G_XAIC], cb[G_F], cb[G_P], T, cb[G_E2], DS_MEM_DEVICE));
        rc = chain_istft(h, h->sub[8], cb[G_E2], T, y_dev, y_bstride); if (rc) return rc;
    }
    const size_t nb = (size_t)B * M * n * 4;
    if (bm_dev && !fused_tail) DS_HIP(h, hipMemcpyAsync(bm_dev, cb[G_BM], nb, hipMemcpyDeviceToDevice, h->stream));   // (the fused tail writes them itself)
    if (al_dev) DS_HIP(h, hipMemcpyAsync(al_dev, cb[G_XA], nb, hipMemcpyDeviceToDevice, h->stream));
    h->al_read[set] = al_dev != nullptr;
    if (p_dev) DS_HIP(h, hipMemcpyAsync(p_dev, cb[G_P], (size_t)B * T * K * 4, hipMemcpyDeviceToDevice, h->stream));
#undef DS_SUB
    rc = flush_tick(h); if (rc) return rc;
    if (h->front_async) {                                   // everything that reads this block's front-end buffers is on the chain's stream by now
        // (or on the blocking-filter branch, ev_fr[8 + set]).  Nothing enqueued there since the tail's fork: that event says the same
        h->fr_mid[set] = join_at_tail && !(al_dev || p_dev);
        if (!h->fr_mid[set]) DS_HIP(h, hipEventRecord(h->ev_fr[2 + set], h->stream));
        h->fr_valid[set] = true;
    }
    return DS_OK;
}


}  // namespace dsi

extern "C" {
int ds_chain_set_aux(ds_handle* h, int which, const float* table, size_t n_floats) {
    if (!h || !table) return fail(h, DS_EINVAL, "ds_chain_set_aux: NULL argument");
    const bool gsc = h->cfg.algo == DS_ALGO_TDGSC || h->cfg.algo == DS_ALGO_FDGSC;
    const bool nb = h->cfg.algo == DS_ALGO_MCSPP_MVDR;
    if (h->cfg.algo != DS_ALGO_SUBBAND_GSC && !gsc && !nb) return fail(h, DS_ESTATE, "ds_chain_set_aux: handle is not a SubbandGSC / TDGSC / FDGSC / McSpp-MVDR chain");
    ds_handle* t = nb ? (which == DS_CHAIN_AUX_COHERENCE ? h->sub[1] : nullptr)
                      : which == DS_CHAIN_AUX_FIR ? h->sub[0] : (which == DS_CHAIN_AUX_COHERENCE && !gsc) ? h->sub[2] : nullptr;
    if (!t) return fail(h, DS_EINVAL, "ds_chain_set_aux: unknown table id");
    { const int jr = set_device(h); if (jr) return jr; }    // stages still running on their own streams come back first
    DS_HIP(h, hipStreamSynchronize(h->stream));
    const int rc = ds_set_aux(t, table, n_floats);
    return rc ? fail(h, rc, t->err) : DS_OK;
}

int ds_mcspp_mvdr_process(ds_handle* h, const float* x, int layout, int n_samples, float* y, float* pp, int mem) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_mcspp_mvdr_process: NULL argument");
    if (h->cfg.algo != DS_ALGO_MCSPP_MVDR) return fail(h, DS_ESTATE, "ds_mcspp_mvdr_process: handle is not a DS_ALGO_MCSPP_MVDR object");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_mcspp_mvdr_process: n_samples must be a multiple of hop");
    if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES) return fail(h, DS_EINVAL, "ds_mcspp_mvdr_process: unknown layout");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, n = n_samples, T = n / h->cfg.hop;
    IoSpec io = {{x, nullptr, nullptr}, {B * M * n * 4, 0, 0}, {y, pp, nullptr, nullptr, nullptr}, {B * n * 4, pp ? B * T * h->K * 4 : 0, 0, 0, 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    rc = nbmvdr_process_device(h, din[0], layout, (long long)(M * n), 0, n_samples, dout[0], (long long)n, pp ? dout[1] : nullptr);
    if (rc) return rc;
    return io_end(h, mem, io, dout);
}

int ds_subband_gsc_process(ds_handle* h, const float* x, int n_samples, float* y, float* fix_output, float* bm_output, float* pp,
                           float* aligned, int mem) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_subband_gsc_process: NULL argument");
    if (h->cfg.algo != DS_ALGO_SUBBAND_GSC) return fail(h, DS_ESTATE, "ds_subband_gsc_process: handle is not a DS_ALGO_SUBBAND_GSC object");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0) return fail(h, DS_ESHAPE, "ds_subband_gsc_process: n_samples must be a multiple of hop");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h); if (rc) return rc;
    const size_t B = h->cfg.batch, M = h->cfg.n_mics, n = n_samples, T = n / h->cfg.hop;
    IoSpec io = {{x, nullptr, nullptr}, {B * M * n * 4, 0, 0}, {y, fix_output, bm_output, pp, aligned},
                 {B * n * 4, fix_output ? B * n * 4 : 0, bm_output ? B * M * n * 4 : 0, pp ? B * T * h->K * 4 : 0, aligned ? B * M * n * 4 : 0}};
    const float* din[3]; float* dout[5];
    rc = io_begin(h, mem, io, din, dout); if (rc) return rc;
    if (mem == DS_MEM_HOST) h->front_open = false;          // the input was just staged on the chain's stream: the front-end stream follows it
    rc = chain2_run(h, din[0], (long long)(M * n), (long long)n, n_samples, dout[0], (long long)n, fix_output ? dout[1] : nullptr,
                    bm_output ? dout[2] : nullptr, pp ? dout[3] : nullptr, aligned ? dout[4] : nullptr);
    if (rc) return rc;
    rc = join_groups(h); if (rc) return rc;                 // the tail's stream, before the outputs are copied back
    return io_end(h, mem, io, dout);
}

}  // extern "C"
