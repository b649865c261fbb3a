// ds_api.hip — handle management and the C-ABI of libdsenh.so (see include/dsenh.h).
// No CPU compute path lives here: every ds_process* call launches the gfx950 kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/dsenh.h"
#include "ds_kernels.hpp"
#include "ds_tables.hpp"

using ds::cf;
using ds::KernelInfo;
using ds::Params;

struct ds_handle {
    ds_config cfg;
    int K, KP, NP, NT;
    KernelInfo ki;
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    // device state
    ds::vec4* bins;
    float* tail_in;
    float* tail_out;
    int* counters;
    ds::vec4* tables;
    cf* steer;
    int steer_per_utt;
    bool steer_set;
    // staging for host-pointer calls
    // cached hipGraph of a ds_process_device_seq() sequence
    hipGraphExec_t graph_exec;
    long long graph_key[16];
    bool graph_valid;
    float* x_stage;
    float* y_stage;
    size_t x_stage_elems, y_stage_elems;
    // params
    int method;
    int mcra_L;
    float alpha_y, alpha_v, diag, gate, mu, out_scale;
    std::string err;
};

namespace {

thread_local std::string g_err;

int fail(ds_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

#define DS_HIP(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, DS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

size_t bins_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * h->NP * h->KP * sizeof(ds::vec4); }
size_t tail_in_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * h->cfg.n_mics * h->cfg.hop * sizeof(float); }
size_t tail_out_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * h->cfg.hop * sizeof(float); }
size_t counters_bytes(const ds_handle* h) { return (size_t)h->cfg.batch * 4 * sizeof(int); }

int set_device(ds_handle* h) {
    DS_HIP(h, hipSetDevice(h->device));
    return DS_OK;
}

int zero_state(ds_handle* h) {
    if (bins_bytes(h)) DS_HIP(h, hipMemsetAsync(h->bins, 0, bins_bytes(h), h->stream));
    DS_HIP(h, hipMemsetAsync(h->tail_in, 0, tail_in_bytes(h), h->stream));
    DS_HIP(h, hipMemsetAsync(h->tail_out, 0, tail_out_bytes(h), h->stream));
    // counters: mcra frm_cnt = 0, ell = 1 (NoiseEstimationBase.py:18,31), spp frm_cnt = 0
    std::vector<int> c((size_t)h->cfg.batch * 4, 0);
    for (int b = 0; b < h->cfg.batch; ++b) c[(size_t)b * 4 + 1] = 1;
    DS_HIP(h, hipMemcpyAsync(h->counters, c.data(), counters_bytes(h), hipMemcpyHostToDevice, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

void fill_params(const ds_handle* h, Params& p) {
    std::memset(&p, 0, sizeof(p));
    p.bins = h->bins;
    p.tail_in = h->tail_in;
    p.tail_out = h->tail_out;
    p.counters = h->counters;
    p.tables = h->tables;
    p.steer = h->steer;
    p.steer_batch_stride = h->steer_per_utt ? (long long)h->K * h->cfg.n_mics : 0;
    p.method = h->method;
    p.mcra_L = h->mcra_L;
    p.out_scale = h->out_scale;
    p.alpha_y = h->alpha_y;
    p.alpha_v = h->alpha_v;
    p.diag = h->diag;
    p.gate = h->gate;
    p.mu = h->mu;
}

}  // namespace

extern "C" {

int ds_version(void) { return DS_VERSION; }

int ds_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* ds_strerror(int code) {
    switch (code) {
        case DS_OK: return "ok";
        case DS_EINVAL: return "invalid argument";
        case DS_ESHAPE: return "shape mismatch (n_samples must be a multiple of hop)";
        case DS_EUNSUPPORTED: return "unsupported configuration (no compiled kernel)";
        case DS_EHIP: return "HIP runtime error";
        case DS_ENOMEM: return "out of memory";
        case DS_ESTATE: return "invalid call order";
        default: return "unknown error";
    }
}

const char* ds_last_error(const ds_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int ds_create(const ds_config* cfg, ds_handle** out) {
    if (!cfg || !out) return fail(nullptr, DS_EINVAL, "ds_create: NULL argument");
    if (cfg->struct_size != (int32_t)sizeof(ds_config)) return fail(nullptr, DS_EINVAL, "ds_create: struct_size mismatch");
    *out = nullptr;
    if (cfg->batch <= 0) return fail(nullptr, DS_EINVAL, "ds_create: batch must be > 0");
    if (cfg->hop * 2 != cfg->nfft)
        return fail(nullptr, DS_EUNSUPPORTED, "ds_create: only hop == nfft/2 is supported");
    KernelInfo ki = {nullptr, 0, 0, 0};
    switch (cfg->algo) {
        case DS_ALGO_FIXED: ki = ds::lookup_fixed(cfg->nfft, cfg->n_mics); break;
        case DS_ALGO_ADAPTIVE:
            ki = cfg->track_ryy ? ds::lookup_adaptive_ryy(cfg->nfft, cfg->n_mics) : ds::lookup_adaptive_noryy(cfg->nfft, cfg->n_mics);
            break;
        case DS_ALGO_GSC: ki = ds::lookup_gsc(cfg->nfft, cfg->n_mics); break;
        default: return fail(nullptr, DS_EINVAL, "ds_create: unknown algo");
    }
    if (!ki.launch) {
        char buf[160];
        snprintf(buf, sizeof buf, "ds_create: no kernel for algo=%d nfft=%d n_mics=%d (nfft in {256,512,1024}, n_mics in {2,4,6,8})",
                 cfg->algo, cfg->nfft, cfg->n_mics);
        return fail(nullptr, DS_EUNSUPPORTED, buf);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, DS_EHIP, "ds_create: no HIP device visible (libdsenh has no CPU path)");
    ds_handle* h = new (std::nothrow) ds_handle();
    if (!h) return fail(nullptr, DS_ENOMEM, "ds_create: host allocation failed");
    h->cfg = *cfg;
    h->ki = ki;
    h->K = cfg->nfft / 2 + 1;
    h->KP = ki.KP;
    h->NP = ki.NP;
    h->NT = ki.NT;
    h->bins = nullptr; h->tail_in = nullptr; h->tail_out = nullptr; h->counters = nullptr;
    h->tables = nullptr; h->steer = nullptr; h->x_stage = nullptr; h->y_stage = nullptr;
    h->x_stage_elems = h->y_stage_elems = 0;
    h->steer_per_utt = 0; h->steer_set = false;
    h->graph_exec = nullptr; h->graph_valid = false;
    h->method = DS_METHOD_MVDR;
    h->mcra_L = cfg->mcra_L > 0 ? cfg->mcra_L : 15;
    h->alpha_y = cfg->alpha_y > 0 ? cfg->alpha_y : 0.8f;
    h->alpha_v = cfg->alpha_v > 0 ? cfg->alpha_v : 0.9998f;
    h->diag = cfg->diag > 0 ? cfg->diag : 1e-6f;
    h->gate = cfg->gate > 0 ? cfg->gate : 0.4f;
    h->mu = cfg->mu > 0 ? cfg->mu : 0.01f;
    if (cfg->device >= 0) h->device = cfg->device;
    else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    h->stream = nullptr; h->ev0 = nullptr; h->ev1 = nullptr;

#define DS_CRE(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            std::string m = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            ds_destroy(h);                                                                       \
            return fail(nullptr, DS_EHIP, m);                                                    \
        }                                                                                        \
    } while (0)

    DS_CRE(hipSetDevice(h->device));
    DS_CRE(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    DS_CRE(hipEventCreate(&h->ev0));
    DS_CRE(hipEventCreate(&h->ev1));
    if (bins_bytes(h)) DS_CRE(hipMalloc((void**)&h->bins, bins_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->tail_in, tail_in_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->tail_out, tail_out_bytes(h)));
    DS_CRE(hipMalloc((void**)&h->counters, counters_bytes(h)));
    const int N = cfg->nfft, NC = N / 2;
    DS_CRE(hipMalloc((void**)&h->steer, (size_t)h->K * cfg->n_mics * sizeof(cf)));
    {
        std::vector<float> blob;
        ds::make_table_blob(N, cfg->hop, blob, h->out_scale);
        DS_CRE(hipMalloc((void**)&h->tables, blob.size() * sizeof(float)));
        DS_CRE(hipMemcpy(h->tables, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    }
#undef DS_CRE
    int rc = zero_state(h);
    if (rc != DS_OK) { std::string m = h->err; ds_destroy(h); return fail(nullptr, rc, m); }
    *out = h;
    return DS_OK;
}

int ds_destroy(ds_handle* h) {
    if (!h) return DS_EINVAL;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->bins); (void)hipFree(h->tail_in); (void)hipFree(h->tail_out); (void)hipFree(h->counters);
    (void)hipFree(h->tables); (void)hipFree(h->steer);
    (void)hipFree(h->x_stage); (void)hipFree(h->y_stage);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return DS_OK;
}

int ds_reset(ds_handle* h) {
    if (!h) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    return zero_state(h);
}

int ds_set_steering(ds_handle* h, const float* steer, int per_utterance) {
    if (!h || !steer) return fail(h, DS_EINVAL, "ds_set_steering: NULL argument");
    int rc = set_device(h);
    if (rc) return rc;
    const size_t one = (size_t)h->K * h->cfg.n_mics * sizeof(cf);
    const size_t need = per_utterance ? one * h->cfg.batch : one;
    if ((per_utterance != 0) != (h->steer_per_utt != 0)) {
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->steer);
        h->steer = nullptr;
        DS_HIP(h, hipMalloc((void**)&h->steer, need));
        h->steer_per_utt = per_utterance ? 1 : 0;
    }
    DS_HIP(h, hipMemcpyAsync(h->steer, steer, need, hipMemcpyHostToDevice, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    h->steer_set = true;
    return DS_OK;
}

int ds_set_param_i(ds_handle* h, int id, int value) {
    if (!h) return DS_EINVAL;
    switch (id) {
        case DS_PARAM_METHOD:
            if (value < 0 || value > 3) return fail(h, DS_EINVAL, "method must be 0..3");
            if (value == DS_METHOD_TFGSC && h->cfg.algo == DS_ALGO_ADAPTIVE && !h->cfg.track_ryy)
                return fail(h, DS_ESTATE, "method TFGSC needs ds_config.track_ryy = 1");
            h->method = value;
            return DS_OK;
        case DS_PARAM_MCRA_L:
            if (value <= 0) return fail(h, DS_EINVAL, "mcra_L must be > 0");
            h->mcra_L = value;
            return DS_OK;
        default: return fail(h, DS_EINVAL, "unknown int parameter id");
    }
}

int ds_set_param_f(ds_handle* h, int id, float value) {
    if (!h) return DS_EINVAL;
    switch (id) {
        case DS_PARAM_ALPHA_Y: h->alpha_y = value; return DS_OK;
        case DS_PARAM_ALPHA_V: h->alpha_v = value; return DS_OK;
        case DS_PARAM_DIAG: h->diag = value; return DS_OK;
        case DS_PARAM_GATE: h->gate = value; return DS_OK;
        case DS_PARAM_MU: h->mu = value; return DS_OK;
        default: return fail(h, DS_EINVAL, "unknown float parameter id");
    }
}

int ds_process_device(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                      int n_samples, float* y_dev, long long y_batch_stride, int first, int count, void* stream) {
    if (!h || !x_dev || !y_dev) return fail(h, DS_EINVAL, "ds_process_device: NULL argument");
    if (!h->steer_set) return fail(h, DS_ESTATE, "ds_process_device: call ds_set_steering first");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0)
        return fail(h, DS_ESHAPE, "ds_process_device: n_samples must be a multiple of hop");
    if (first < 0 || count < 0 || first + count > h->cfg.batch)
        return fail(h, DS_EINVAL, "ds_process_device: utterance range outside the handle's batch");
    if (layout != DS_LAYOUT_SAMPLES_CHANNELS && layout != DS_LAYOUT_CHANNELS_SAMPLES)
        return fail(h, DS_EINVAL, "ds_process_device: unknown layout");
    if (((uintptr_t)x_dev & 15) || ((uintptr_t)y_dev & 15) || (x_batch_stride & 3) || (x_chan_stride & 3))
        return fail(h, DS_EINVAL, "ds_process_device: device buffers must be 16-byte aligned");
    if (n_samples == 0 || count == 0) return DS_OK;
    int rc = set_device(h);
    if (rc) return rc;
    Params p;
    fill_params(h, p);
    p.x = x_dev;
    p.y = y_dev;
    p.x_batch_stride = x_batch_stride;
    p.y_batch_stride = y_batch_stride;
    if (layout == DS_LAYOUT_CHANNELS_SAMPLES) { p.x_sample_stride = 1; p.x_chan_stride = x_chan_stride > 0 ? x_chan_stride : n_samples; }
    else { p.x_sample_stride = h->cfg.n_mics; p.x_chan_stride = 1; }
    p.T = n_samples / h->cfg.hop;
    p.batch0 = first;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    DS_HIP(h, h->ki.launch(p, count, s));
    return DS_OK;
}

int ds_process_device_seq(ds_handle* h, const float* x_dev, int layout, long long x_batch_stride, long long x_chan_stride,
                          long long x_call_stride, int n_samples_per_call, int n_calls, float* y_dev,
                          long long y_batch_stride, long long y_call_stride, int first, int count, void* stream,
                          int graph) {
    if (!h) return DS_EINVAL;
    if (n_calls < 0 || (x_call_stride & 3) || (y_call_stride & 3))
        return fail(h, DS_EINVAL, "ds_process_device_seq: bad n_calls / call strides (must be multiples of 4 elements)");
    if (n_calls == 0) return DS_OK;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    if (graph == 0) {
        for (int i = 0; i < n_calls; ++i) {
            int rc = ds_process_device(h, x_dev + (long long)i * x_call_stride, layout, x_batch_stride, x_chan_stride,
                                       n_samples_per_call, y_dev + (long long)i * y_call_stride, y_batch_stride, first,
                                       count, (void*)s);
            if (rc) return rc;
        }
        return DS_OK;
    }
    int rc = set_device(h);
    if (rc) return rc;
    float fl[6] = {h->alpha_y, h->alpha_v, h->diag, h->gate, h->mu, 0.0f};
    long long fbits[3];
    std::memcpy(fbits, fl, sizeof fbits);
    const long long key[16] = {(long long)(uintptr_t)x_dev, (long long)(uintptr_t)y_dev, layout, x_batch_stride, x_chan_stride,
                               x_call_stride, n_samples_per_call, n_calls, y_batch_stride, y_call_stride,
                               ((long long)first << 32) | (unsigned)count, ((long long)h->method << 32) | (unsigned)h->mcra_L,
                               fbits[0], fbits[1], fbits[2], (long long)(uintptr_t)h->steer};
    if (!h->graph_valid || std::memcmp(key, h->graph_key, sizeof key) != 0) {
        if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
        h->graph_valid = false;
        hipStream_t cs = h->stream;                       // capture on the handle's own stream
        DS_HIP(h, hipStreamSynchronize(cs));
        DS_HIP(h, hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        int crc = DS_OK;
        for (int i = 0; i < n_calls && crc == DS_OK; ++i)
            crc = ds_process_device(h, x_dev + (long long)i * x_call_stride, layout, x_batch_stride, x_chan_stride,
                                    n_samples_per_call, y_dev + (long long)i * y_call_stride, y_batch_stride, first, count,
                                    (void*)cs);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(cs, &g);
        if (crc != DS_OK) { if (g) (void)hipGraphDestroy(g); return crc; }
        if (e != hipSuccess) return fail(h, DS_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return fail(h, DS_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        std::memcpy(h->graph_key, key, sizeof key);
        h->graph_valid = true;
    }
    if (graph == 2) return DS_OK;
    DS_HIP(h, hipGraphLaunch(h->graph_exec, s));
    return DS_OK;
}

int ds_process(ds_handle* h, const float* x, int layout, int n_samples, float* y) {
    if (!h || !x || !y) return fail(h, DS_EINVAL, "ds_process: NULL argument");
    if (n_samples < 0 || n_samples % h->cfg.hop != 0)
        return fail(h, DS_ESHAPE, "ds_process: n_samples must be a multiple of hop");
    if (n_samples == 0) return DS_OK;
    int rc = set_device(h);
    if (rc) return rc;
    const size_t B = (size_t)h->cfg.batch, M = (size_t)h->cfg.n_mics;
    const size_t xe = B * M * (size_t)n_samples, ye = B * (size_t)n_samples;
    if (xe > h->x_stage_elems) {
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->x_stage); h->x_stage = nullptr; h->x_stage_elems = 0;
        DS_HIP(h, hipMalloc((void**)&h->x_stage, xe * sizeof(float)));
        h->x_stage_elems = xe;
    }
    if (ye > h->y_stage_elems) {
        DS_HIP(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->y_stage); h->y_stage = nullptr; h->y_stage_elems = 0;
        DS_HIP(h, hipMalloc((void**)&h->y_stage, ye * sizeof(float)));
        h->y_stage_elems = ye;
    }
    DS_HIP(h, hipMemcpyAsync(h->x_stage, x, xe * sizeof(float), hipMemcpyHostToDevice, h->stream));
    rc = ds_process_device(h, h->x_stage, layout, (long long)(M * (size_t)n_samples), 0, n_samples, h->y_stage,
                           (long long)n_samples, 0, h->cfg.batch, nullptr);
    if (rc) return rc;
    DS_HIP(h, hipMemcpyAsync(y, h->y_stage, ye * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_synchronize(ds_handle* h) {
    if (!h) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    return DS_OK;
}

int ds_timing_begin(ds_handle* h) {
    if (!h) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipEventRecord(h->ev0, h->stream));
    return DS_OK;
}

int ds_timing_end(ds_handle* h, float* elapsed_ms) {
    if (!h || !elapsed_ms) return DS_EINVAL;
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipEventRecord(h->ev1, h->stream));
    DS_HIP(h, hipEventSynchronize(h->ev1));
    DS_HIP(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    return DS_OK;
}

size_t ds_field_bytes(const ds_handle* h, int field) {
    if (!h) return 0;
    const size_t B = h->cfg.batch, K = h->K, M = h->cfg.n_mics;
    const bool ad = h->cfg.algo == DS_ALGO_ADAPTIVE, gsc = h->cfg.algo == DS_ALGO_GSC;
    switch (field) {
        case DS_FIELD_RVV: return ad ? B * K * M * M * 2 * sizeof(float) : 0;
        case DS_FIELD_RYY: return (ad && h->cfg.track_ryy) ? B * K * M * M * 2 * sizeof(float) : 0;
        case DS_FIELD_MCRA_S: case DS_FIELD_MCRA_SMIN: case DS_FIELD_MCRA_STMP: case DS_FIELD_MCRA_P:
        case DS_FIELD_MCRA_LAMBDA_D: return ad ? B * K * sizeof(float) : 0;
        case DS_FIELD_PHI_YY: case DS_FIELD_PHI_VV: return gsc ? B * K * M * M * sizeof(float) : 0;
        case DS_FIELD_G_AIC: return gsc ? B * K * (M - 1) * 2 * sizeof(float) : 0;
        case DS_FIELD_STFT_TAIL: return tail_in_bytes(h);
        case DS_FIELD_OLA_TAIL: return tail_out_bytes(h);
        case DS_FIELD_COUNTERS: return counters_bytes(h);
        default: return 0;
    }
}

int ds_get_state(ds_handle* h, int field, void* dst, size_t bytes) {
    if (!h || !dst) return fail(h, DS_EINVAL, "ds_get_state: NULL argument");
    const size_t need = ds_field_bytes(h, field);
    if (need == 0) return fail(h, DS_EINVAL, "ds_get_state: field not available for this algo/config");
    if (bytes != need) return fail(h, DS_ESHAPE, "ds_get_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    if (field == DS_FIELD_STFT_TAIL) { DS_HIP(h, hipMemcpy(dst, h->tail_in, need, hipMemcpyDeviceToHost)); return DS_OK; }
    if (field == DS_FIELD_OLA_TAIL) { DS_HIP(h, hipMemcpy(dst, h->tail_out, need, hipMemcpyDeviceToHost)); return DS_OK; }
    if (field == DS_FIELD_COUNTERS) { DS_HIP(h, hipMemcpy(dst, h->counters, need, hipMemcpyDeviceToHost)); return DS_OK; }
    // per-bin fields: pull the raw planes and unpack on the host
    std::vector<float> raw(bins_bytes(h) / sizeof(float));
    DS_HIP(h, hipMemcpy(raw.data(), h->bins, bins_bytes(h), hipMemcpyDeviceToHost));
    const int B = h->cfg.batch, K = h->K, M = h->cfg.n_mics, KP = h->KP, NP = h->NP;
    auto at = [&](int b, int k, int f) -> float {   // float f of bin k of utterance b
        return raw[(((size_t)b * NP + f / 4) * KP + k) * 4 + (f % 4)];
    };
    float* out = (float*)dst;
    auto herm = [&](int d0, int o0) {
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < M; ++i)
                    for (int j = 0; j < M; ++j) {
                        float re, im;
                        if (i == j) { re = at(b, k, d0 + i); im = 0.0f; }
                        else if (i < j) { int q = ds::off_index(i, j, M); re = at(b, k, o0 + 2 * q); im = at(b, k, o0 + 2 * q + 1); }
                        else { int q = ds::off_index(j, i, M); re = at(b, k, o0 + 2 * q); im = -at(b, k, o0 + 2 * q + 1); }
                        float* o = out + ((((size_t)b * K + k) * M + i) * M + j) * 2;
                        o[0] = re; o[1] = im;
                    }
    };
    auto scalar = [&](int f) {
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k) out[(size_t)b * K + k] = at(b, k, f);
    };
    auto sym = [&](int s0) {
        for (int b = 0; b < B; ++b)
            for (int k = 0; k < K; ++k)
                for (int i = 0; i < M; ++i)
                    for (int j = 0; j < M; ++j) {
                        int q = i <= j ? ds::sym_index(i, j, M) : ds::sym_index(j, i, M);
                        out[(((size_t)b * K + k) * M + i) * M + j] = at(b, k, s0 + q);
                    }
    };
    const int MCS = M * M;
    switch (field) {
        case DS_FIELD_RVV: herm(0, M); break;
        case DS_FIELD_RYY: herm(M * M + 5, M * M + 5 + M); break;
        case DS_FIELD_MCRA_S: scalar(MCS + 0); break;
        case DS_FIELD_MCRA_SMIN: scalar(MCS + 1); break;
        case DS_FIELD_MCRA_STMP: scalar(MCS + 2); break;
        case DS_FIELD_MCRA_P: scalar(MCS + 3); break;
        case DS_FIELD_MCRA_LAMBDA_D: scalar(MCS + 4); break;
        case DS_FIELD_PHI_YY: sym(0); break;
        case DS_FIELD_PHI_VV: sym(M * (M + 1) / 2); break;
        case DS_FIELD_G_AIC:
            for (int b = 0; b < B; ++b)
                for (int k = 0; k < K; ++k)
                    for (int i = 0; i < 2 * (M - 1); ++i)
                        out[((size_t)b * K + k) * 2 * (M - 1) + i] = at(b, k, M * (M + 1) + i);
            break;
        default: return fail(h, DS_EINVAL, "ds_get_state: unknown field");
    }
    return DS_OK;
}

size_t ds_state_bytes(const ds_handle* h) {
    if (!h) return 0;
    return bins_bytes(h) + tail_in_bytes(h) + tail_out_bytes(h) + counters_bytes(h);
}

int ds_export_state(ds_handle* h, void* dst, size_t bytes) {
    if (!h || !dst) return fail(h, DS_EINVAL, "ds_export_state: NULL argument");
    if (bytes != ds_state_bytes(h)) return fail(h, DS_ESHAPE, "ds_export_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    char* d = (char*)dst;
    if (bins_bytes(h)) DS_HIP(h, hipMemcpy(d, h->bins, bins_bytes(h), hipMemcpyDeviceToHost));
    d += bins_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->tail_in, tail_in_bytes(h), hipMemcpyDeviceToHost)); d += tail_in_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->tail_out, tail_out_bytes(h), hipMemcpyDeviceToHost)); d += tail_out_bytes(h);
    DS_HIP(h, hipMemcpy(d, h->counters, counters_bytes(h), hipMemcpyDeviceToHost));
    return DS_OK;
}

int ds_import_state(ds_handle* h, const void* src, size_t bytes) {
    if (!h || !src) return fail(h, DS_EINVAL, "ds_import_state: NULL argument");
    if (bytes != ds_state_bytes(h)) return fail(h, DS_ESHAPE, "ds_import_state: byte size mismatch");
    int rc = set_device(h);
    if (rc) return rc;
    DS_HIP(h, hipStreamSynchronize(h->stream));
    const char* s = (const char*)src;
    if (bins_bytes(h)) DS_HIP(h, hipMemcpy(h->bins, s, bins_bytes(h), hipMemcpyHostToDevice));
    s += bins_bytes(h);
    DS_HIP(h, hipMemcpy(h->tail_in, s, tail_in_bytes(h), hipMemcpyHostToDevice)); s += tail_in_bytes(h);
    DS_HIP(h, hipMemcpy(h->tail_out, s, tail_out_bytes(h), hipMemcpyHostToDevice)); s += tail_out_bytes(h);
    DS_HIP(h, hipMemcpy(h->counters, s, counters_bytes(h), hipMemcpyHostToDevice));
    return DS_OK;
}

}  // extern "C"
