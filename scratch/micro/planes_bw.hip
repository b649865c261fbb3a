// planes_bw.hip — what HBM rate does a "read the per-bin state, compute, write it back" kernel get as a function of the state layout
// (4-byte planes [B][NF][KP] vs 16-byte planes [B][NF/4][KP] float4), the occupancy (waves per SIMD) and the cache policy?
// One thread per (utterance, bin); NF floats of state per bin; FMA chain of `work` dependent steps between load and store.
// Build: hipcc -O3 --offload-arch=gfx950 planes_bw.hip -o planes_bw ; run: ./planes_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#ifndef KPV
#define KPV 260          // -DKPV=264: plane rows on 128-byte lines (16-byte planes: 264 x 16 B = 33 lines; 260 x 16 B = 32.5)
#endif
constexpr int NF = 72, KP = KPV, K = 257;
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int W, bool NT> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
k_planes4(float* st, int B, int work) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = (int)(i / KP), k = (int)(i - (long long)b * KP);
    if (b >= B || k >= K) return;
    float* base = st + (long long)b * NF * KP + k;
    float v[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) v[f] = NT ? __builtin_nontemporal_load(base + f * KP) : base[f * KP];
    float acc = v[0];
    for (int w = 0; w < work; ++w) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
#pragma unroll
    for (int f = 0; f < NF; ++f) { const float o = v[f] + acc * 1e-30f; if (NT) __builtin_nontemporal_store(o, base + f * KP); else base[f * KP] = o; }
}

template <int W, bool NT> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
k_planes16(vf4* st, int B, int work) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = (int)(i / KP), k = (int)(i - (long long)b * KP);
    if (b >= B || k >= K) return;
    vf4* base = st + (long long)b * (NF / 4) * KP + k;
    vf4 v[NF / 4];
#pragma unroll
    for (int f = 0; f < NF / 4; ++f) v[f] = NT ? __builtin_nontemporal_load(base + f * KP) : base[f * KP];
    float acc = v[0].x;
    for (int w = 0; w < work; ++w) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
#pragma unroll
    for (int f = 0; f < NF / 4; ++f) {
        vf4 o = v[f]; o.x += acc * 1e-30f;
        if (NT) __builtin_nontemporal_store(o, base + f * KP); else base[f * KP] = o;
    }
}


typedef float vf2 __attribute__((ext_vector_type(2)));
template <int W, bool NT> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
k_planes8(vf2* st, int B, int work) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = (int)(i / KP), k = (int)(i - (long long)b * KP);
    if (b >= B || k >= K) return;
    vf2* base = st + (long long)b * (NF / 2) * KP + k;
    vf2 v[NF / 2];
#pragma unroll
    for (int f = 0; f < NF / 2; ++f) v[f] = NT ? __builtin_nontemporal_load(base + f * KP) : base[f * KP];
    float acc = v[0].x;
    for (int w = 0; w < work; ++w) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
#pragma unroll
    for (int f = 0; f < NF / 2; ++f) {
        vf2 o = v[f]; o.x += acc * 1e-30f;
        if (NT) __builtin_nontemporal_store(o, base + f * KP); else base[f * KP] = o;
    }
}
// 16-byte planes, plain loads, non-temporal stores (and the reverse)
template <int W, bool NTL, bool NTS> __global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
k_planes16m(vf4* st, int B, int work) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = (int)(i / KP), k = (int)(i - (long long)b * KP);
    if (b >= B || k >= K) return;
    vf4* base = st + (long long)b * (NF / 4) * KP + k;
    vf4 v[NF / 4];
#pragma unroll
    for (int f = 0; f < NF / 4; ++f) v[f] = NTL ? __builtin_nontemporal_load(base + f * KP) : base[f * KP];
    float acc = v[0].x;
    for (int w = 0; w < work; ++w) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
#pragma unroll
    for (int f = 0; f < NF / 4; ++f) {
        vf4 o = v[f]; o.x += acc * 1e-30f;
        if (NTS) __builtin_nontemporal_store(o, base + f * KP); else base[f * KP] = o;
    }
}

template <class F> static float time_it(F launch, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int Bs[3] = {1024, 2048, 8192};
    for (int bi = 0; bi < 3; ++bi) {
        const int B = Bs[bi];
        const size_t n = (size_t)B * NF * KP;
        float* st; CK(hipMalloc(&st, n * 4)); CK(hipMemset(st, 0, n * 4));
        const int blocks = (int)(((long long)B * KP + 255) / 256);
        const double bytes = 2.0 * B * K * NF * 4;
        for (int work : {0, 1300}) {
            printf("B = %d (%.0f MB moved per launch), %d dependent FMAs between load and store\n", B, bytes / 1e6, work);
#define RUN(name, kern) { float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (decltype(st))st, B, work); }, 20); \
            printf("  %-34s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); }
#define RUN16(name, kern) { float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (vf4*)st, B, work); }, 20); \
            printf("  %-34s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); }
            RUN("4-byte planes, 2 waves/SIMD", (k_planes4<2, false>));
            RUN("4-byte planes, 2 waves/SIMD, nt", (k_planes4<2, true>));
            RUN("4-byte planes, 4 waves/SIMD", (k_planes4<4, false>));
            RUN("4-byte planes, 4 waves/SIMD, nt", (k_planes4<4, true>));
#define RUN8(name, kern) { float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, (vf2*)st, B, work); }, 20); \
            printf("  %-34s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); }
            RUN8("8-byte planes, 2 waves/SIMD", (k_planes8<2, false>));
            RUN8("8-byte planes, 2 waves/SIMD, nt", (k_planes8<2, true>));
            RUN8("8-byte planes, 4 waves/SIMD", (k_planes8<4, false>));
            RUN16("16-byte planes, 2 waves/SIMD", (k_planes16<2, false>));
            RUN16("16-byte, 2 w/SIMD, nt stores only", (k_planes16m<2, false, true>));
            RUN16("16-byte, 2 w/SIMD, nt loads only", (k_planes16m<2, true, false>));
            RUN16("16-byte planes, 2 waves/SIMD, nt", (k_planes16<2, true>));
            RUN16("16-byte planes, 4 waves/SIMD", (k_planes16<4, false>));
            RUN16("16-byte planes, 4 waves/SIMD, nt", (k_planes16<4, true>));
        }
        CK(hipFree(st));
    }
    return 0;
}
