// valu_rate.hip — issue cost of the vector instructions the frame kernels are made of, per SIMD, at 1 / 2 / 4 waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 scratch/micro/valu_rate.hip -o scratch/micro/valu_rate && scratch/micro/valu_rate
// Every wave of a 256-thread block (one wave per SIMD) runs REPS x 32 independent instructions of one kind (32 accumulators: no dependent
// chain shorter than 32 instructions).  blocks = 256 CUs x waves-per-SIMD, all resident at once.  Reported per kind:
//   ns per instruction per SIMD  = kernel wall time / (REPS * 32 * waves per SIMD)          (hipEvents around a second, warm launch)
//   shader clock                 = delta s_memtime / delta s_memrealtime * 100 MHz            (median wave)
//   cycles per instruction per SIMD = the product
// Under rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE the same launches calibrate the SQ counters:
// how many quad-cycles of SQ_ACTIVE_INST_VALU one instruction of each kind is booked as.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));
enum { OP_FMA, OP_FMAC, OP_MUL, OP_ADD, OP_MOV, OP_PK_FMA, OP_PK_FMA_SEL, OP_PK_MUL, OP_PK_ADD, OP_RCP, OP_CNDMASK, OP_MIX_PK_ADD, OP_MIX_PK_FMA,
       OP_LDS_R64, OP_LDS_W64, OP_FMA64, OP_MUL64, OP_ADD64, OP_CVT64, N_OPS };
static const char* names[] = {"v_fma_f32 (3 distinct srcs)", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_mov_b32", "v_pk_fma_f32", "v_pk_fma_f32 op_sel/neg",
                              "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "v_cndmask_b32 (sgpr mask)", "v_pk_fma + v_add (per pair)",
                              "v_pk_fma + v_fma (per pair)", "ds_read_b64", "ds_write_b64",
                              "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64_f32"};

template <int OP> __global__ void __launch_bounds__(256) rate(long long* out, int reps, float seed) {
    __shared__ f2 lds[256 * 9];
    f2 a[16];
    float s[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i].x = seed + i + threadIdx.x; a[i].y = seed - i; }
#pragma unroll
    for (int i = 0; i < 32; ++i) s[i] = seed * i + threadIdx.x;
    f2 b = {seed * 0.5f, seed * 0.25f}, b2 = {seed * 0.75f, seed * 0.35f};
    float c = seed * 0.125f, c2 = seed * 0.375f;
    lds[threadIdx.x] = b;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        if constexpr (OP == OP_FMA) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(c), "v"(c2));
        } else if constexpr (OP == OP_FMAC) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(c), "v"(c2));
        } else if constexpr (OP == OP_MUL) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
        } else if constexpr (OP == OP_ADD) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c));
        } else if constexpr (OP == OP_MOV) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(c));
        } else if constexpr (OP == OP_PK_FMA) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(b2));
        } else if constexpr (OP == OP_PK_FMA_SEL) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(b2));
        } else if constexpr (OP == OP_PK_MUL) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if constexpr (OP == OP_PK_ADD) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if constexpr (OP == OP_RCP) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
        } else if constexpr (OP == OP_CNDMASK) {
            unsigned long long m = 0x5555aaaa5555aaaaull;
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c), "s"(m));
        } else if constexpr (OP == OP_MIX_PK_ADD) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(b2));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i + 16 * j]) : "v"(c));
                }
        } else if constexpr (OP == OP_MIX_PK_FMA) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(b2));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i + 16 * j]) : "v"(c), "v"(c2));
                }
        } else if constexpr (OP == OP_FMA64) {                // the notebook-MVDR operator's estimation core and eigen-solver run in double (ds_linalg64.hpp)
            double cd_ = (double)c, cd2 = (double)c2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(*reinterpret_cast<double*>(&a[i])) : "v"(cd_), "v"(cd2));
        } else if constexpr (OP == OP_MUL64) {
            double cd_ = (double)c;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&a[i])) : "v"(cd_));
        } else if constexpr (OP == OP_ADD64) {
            double cd_ = (double)c;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&a[i])) : "v"(cd_));
        } else if constexpr (OP == OP_CVT64) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*reinterpret_cast<double*>(&a[i])) : "v"(s[i + 16 * j]));
        } else if constexpr (OP == OP_LDS_R64) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] += lds[threadIdx.x + 256 * (i & 7)];
            asm volatile("" ::: "memory");
        } else if constexpr (OP == OP_LDS_W64) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) lds[threadIdx.x + 256 * (i & 7)] = a[i];
            asm volatile("" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float acc = c;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += a[i].x + a[i].y;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += s[i];
    if (acc == 12345.678f) out[0] = 0;                       // keeps the chains alive
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[1 + 2 * w] = t1 - t0;
        out[2 + 2 * w] = r1 - r0;
    }
}

template <int OP> void run(long long* d_out, std::vector<long long>& h) {
    const int reps = 20000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        rate<OP><<<blocks, 256>>>(d_out, reps, 1.0f);            // warm
        (void)hipEventRecord(e0);
        rate<OP><<<blocks, 256>>>(d_out, reps, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), d_out, (1 + blocks * 8) * sizeof(long long), hipMemcpyDeviceToHost);
        std::vector<double> clk;
        for (int w = 0; w < blocks * 4; ++w) clk.push_back((double)h[1 + 2 * w] / (double)h[2 + 2 * w] * 0.1);   // GHz
        std::sort(clk.begin(), clk.end());
        const double ghz = clk[clk.size() / 2];
        const double per = (OP == OP_MIX_PK_ADD || OP == OP_MIX_PK_FMA) ? 32.0 : 32.0;    // instructions (or pairs) per rep
        const double ns = ms * 1e6 / (reps * per * wps);
        printf("%-30s %d waves/SIMD: %6.3f ns per instruction per SIMD, clock %.2f GHz => %5.2f cycles   (kernel %.3f ms)\n",
               names[OP], wps, ns, ghz, ns * ghz, ms);
    }
}

int main() {
    long long* d_out;
    (void)hipMalloc(&d_out, (1 + 1024 * 8) * sizeof(long long));
    std::vector<long long> h(1 + 1024 * 8);
    run<OP_FMA>(d_out, h); run<OP_FMAC>(d_out, h); run<OP_MUL>(d_out, h); run<OP_ADD>(d_out, h); run<OP_MOV>(d_out, h);
    run<OP_PK_FMA>(d_out, h); run<OP_PK_FMA_SEL>(d_out, h); run<OP_PK_MUL>(d_out, h); run<OP_PK_ADD>(d_out, h);
    run<OP_RCP>(d_out, h); run<OP_CNDMASK>(d_out, h); run<OP_MIX_PK_ADD>(d_out, h); run<OP_MIX_PK_FMA>(d_out, h);
    run<OP_LDS_R64>(d_out, h); run<OP_LDS_W64>(d_out, h);
    run<OP_FMA64>(d_out, h); run<OP_MUL64>(d_out, h); run<OP_ADD64>(d_out, h); run<OP_CVT64>(d_out, h);
    return 0;
}
