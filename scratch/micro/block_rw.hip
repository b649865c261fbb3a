// block_rw.hip — what HBM gives the wide-tap WPE kernel's ACCESS PATTERN with no arithmetic in it: one wavefront per block of BLK bytes (the
// state of one bin: 29 KB at C N = 80), read in CH-byte chunks through LDS-DMA into an LDS tile and written back from the tile, chunk
// after chunk; LDSB bytes of LDS per wavefront set how many wavefronts a CU holds (160 KB / LDSB, at most 8 per SIMD).
// hipcc -O3 --offload-arch=gfx950 scratch/micro/block_rw.hip -o scratch/micro/block_rw && scratch/micro/block_rw
// Reported: (bytes read + bytes written) / kernel time — the number the kernel's roofline fraction is measured against 8 TB/s with.
//   mode 0: read chunk, wait, write chunk back (the kernel's order with the frame loop taken out: chunks one after the other)
//   mode 1: read every chunk of the block first (tile = the whole block), one wait, then write all
//   mode 2: mode 0, and SPIN dependent v_fma per chunk between the read and the write (a stand-in for gather + update + scatter)
// and the variants: nontemporal loads / stores, a second array to write to (ping-pong state), read only, write only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// AUXL: cache-policy bits of the loads (0, or 2 = nt); NTS: nontemporal stores; OOP: write to a second array (ping-pong state)
// WR: 0 = read + write, 1 = read only, 2 = write only
template <int MODE, int AUXL = 0, bool NTS = false, bool OOP = false, int WR = 0>
__global__ void __launch_bounds__(64) rw(float* state, float* state2, long long blk_floats, int chunk_floats, int spin, float seed) {
    extern __shared__ float4 tile[];
    const int l = threadIdx.x;
    float* st = state + (long long)blockIdx.x * blk_floats;
    float* so = (OOP ? state2 : state) + (long long)blockIdx.x * blk_floats;
    auto put = [&](float* dst, float4 v) {
        if constexpr (WR == 1) { if (v.x == 12345.678f) *reinterpret_cast<float4*>(dst) = v; }
        else if constexpr (NTS) { typedef float v4 __attribute__((ext_vector_type(4))); v4 u = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(u, reinterpret_cast<v4*>(dst)); }
        else *reinterpret_cast<float4*>(dst) = v;
    };
    const int nch = (int)((blk_floats + chunk_floats - 1) / chunk_floats);
    float acc = seed + l;
    if constexpr (MODE == 1) {
        for (int w = 0; w < blk_floats; w += 256)
            if (w + 4 * l < blk_floats)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st + w + 4 * l), (__attribute__((address_space(3))) void*)(tile + w / 4), 16, 0, AUXL);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < spin; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc) : "v"(seed));
        if (acc == 12345.678f) tile[l].x = acc;
        for (int w = 0; w < blk_floats; w += 256)
            if (w + 4 * l < blk_floats) put(so + w + 4 * l, tile[w / 4 + l]);
    } else {
        for (int c = 0; c < nch; ++c) {
            const long long w0 = (long long)c * chunk_floats;
            const long long w1 = w0 + chunk_floats < blk_floats ? w0 + chunk_floats : blk_floats;
            for (long long w = w0; w < w1; w += 256)
                if (WR != 2 && w + 4 * l < w1)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st + w + 4 * l), (__attribute__((address_space(3))) void*)(tile + (w - w0) / 4), 16, 0, AUXL);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __builtin_amdgcn_wave_barrier();
            if constexpr (MODE == 2) {
                for (int i = 0; i < spin; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc) : "v"(seed));
                if (acc == 12345.678f) tile[l].x = acc;
            }
            for (long long w = w0; w < w1; w += 256)
                if (w + 4 * l < w1) put(so + w + 4 * l, tile[(w - w0) / 4 + l]);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// mode 3: a workgroup of WV wavefronts, one block each, the read and the write phases of its wavefronts aligned by workgroup barriers
// (chunks through LDS as in mode 0; LDS per workgroup = WV tiles)
template <int WV, int AUXL, bool NTS>
__global__ void __launch_bounds__(64 * WV) rw_aligned(float* state, float*, long long blk_floats, int chunk_floats, int, float) {
    extern __shared__ float4 tile[];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float4* my = tile + (size_t)wv * (chunk_floats / 4);
    float* st = state + ((long long)blockIdx.x * WV + wv) * blk_floats;
    const int nch = (int)((blk_floats + chunk_floats - 1) / chunk_floats);
    for (int c = 0; c < nch; ++c) {
        const long long w0 = (long long)c * chunk_floats;
        const long long w1 = w0 + chunk_floats < blk_floats ? w0 + chunk_floats : blk_floats;
        __syncthreads();
        for (long long w = w0; w < w1; w += 256)
            if (w + 4 * l < w1)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(st + w + 4 * l), (__attribute__((address_space(3))) void*)(my + (w - w0) / 4), 16, 0, AUXL);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        for (long long w = w0; w < w1; w += 256)
            if (w + 4 * l < w1) {
                const float4 v = my[(w - w0) / 4 + l];
                if constexpr (NTS) { typedef float v4 __attribute__((ext_vector_type(4))); v4 u = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(u, reinterpret_cast<v4*>(st + w + 4 * l)); }
                else *reinterpret_cast<float4*>(st + w + 4 * l) = v;
            }
    }
}
// mode 4: the frame kernels' pattern: a 256-thread workgroup reads NB consecutive blocks (NB x 29 KB) into REGISTERS in one burst of
// 16-byte loads, one barrier, and writes them back in one burst
template <int NV, bool NT>
__global__ void __launch_bounds__(256) rw_burst(float* state, float*, long long blk_floats, int, int, float) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const long long burst = (long long)NV * 256 * 4;                                     // floats per workgroup
    v4* st = reinterpret_cast<v4*>(state + (long long)blockIdx.x * burst);
    v4 r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = NT ? __builtin_nontemporal_load(st + i * 256 + threadIdx.x) : st[i * 256 + threadIdx.x];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) { if (NT) __builtin_nontemporal_store(r[i], st + i * 256 + threadIdx.x); else st[i * 256 + threadIdx.x] = r[i]; }
}

// mode 5: mode 4 with SPIN dependent FMAs between the burst of loads and the burst of stores (the frame kernels' shape: read the state,
// run the hop, write the state), dynamic LDS only to set the number of resident workgroups per CU
template <int NV, bool NT>
__global__ void __launch_bounds__(256) rw_burst_spin(float* state, float*, long long, int, int spin, float seed) {
    extern __shared__ float4 tile[];
    typedef float v4 __attribute__((ext_vector_type(4)));
    const long long burst = (long long)NV * 256 * 4;
    v4* st = reinterpret_cast<v4*>(state + (long long)blockIdx.x * burst);
    v4 r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = NT ? __builtin_nontemporal_load(st + i * 256 + threadIdx.x) : st[i * 256 + threadIdx.x];
    float acc = r[0].x + seed;
    for (int i = 0; i < spin; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc) : "v"(seed));
    if (acc == 12345.678f) tile[threadIdx.x].x = acc;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) { if (NT) __builtin_nontemporal_store(r[i], st + i * 256 + threadIdx.x); else st[i * 256 + threadIdx.x] = r[i]; }
}

// mode 6: the NARROW WPE kernel's shape (C N = 16): a wavefront holds four consecutive bins, 16 lanes each; every instruction moves one
// 256-byte piece per bin (16 lanes x 16 B), i.e. four pieces SB bytes apart; pieces go straight to registers and back (no LDS).
// SB = bytes per bin: 2244 (the kernel's block, pieces straddle 128-byte lines) or 2304 (padded to whole lines)
template <bool NT>
__global__ void __launch_bounds__(256) rw_narrow(float* state, float*, long long sb_floats, int, int, float) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 15, bin = threadIdx.x >> 4;                       // 16 bins per 256-thread workgroup
    float* st = state + ((long long)blockIdx.x * 16 + bin) * sb_floats;
    const int pieces = (int)(sb_floats * 4 / 256);                                   // whole 256-byte pieces of the bin's block (8 of them at 2244 / 9 at 2304)
    v4 r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i)
        if (i < pieces) r[i] = NT ? __builtin_nontemporal_load(reinterpret_cast<v4*>(st + i * 64 + lane * 4)) : *reinterpret_cast<v4*>(st + i * 64 + lane * 4);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 9; ++i)
        if (i < pieces) { if (NT) __builtin_nontemporal_store(r[i], reinterpret_cast<v4*>(st + i * 64 + lane * 4)); else *reinterpret_cast<v4*>(st + i * 64 + lane * 4) = r[i]; }
}

int main(int argc, char** argv) {
    const long long blocks = argc > 1 ? atoll(argv[1]) : 132096;          // wpe_nb: 1024 utterances x 129 bins
    const long long blk_bytes = (argc > 3 && argv[2][0] == 'b') ? atoll(argv[3]) : 29120;     // wpe_bin_floats(4, 20) * 4 rounded to 16 B; `block_rw <blocks> blk <bytes>`: another block size
    const long long blk_floats = blk_bytes / 4;
    float *state, *state2;
    CK(hipMalloc(&state, blocks * blk_bytes));
    CK(hipMalloc(&state2, blocks * blk_bytes));
    CK(hipMemset(state, 0, blocks * blk_bytes));
    CK(hipMemset(state2, 0, blocks * blk_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    typedef void (*K)(float*, float*, long long, int, int, float);
    struct Case { const char* name; K k; int chunk_bytes; int lds_bytes; int spin; double traffic; };
    std::vector<Case> cases = {
        {"chunks, in place", rw<0>, 13312, 20480, 0, 2}, {"chunks, in place", rw<0>, 13312, 14336, 0, 2},
        {"chunks, in place", rw<0>, 6656, 20480, 0, 2}, {"chunks, in place", rw<0>, 6656, 6656, 0, 2}, {"chunks, in place", rw<0>, 4096, 5120, 0, 2},
        {"whole block, in place", rw<1>, 29120, 32768, 0, 2},
        {"chunks, nt loads", rw<0, 2>, 13312, 20480, 0, 2}, {"chunks, nt stores", rw<0, 0, true>, 13312, 20480, 0, 2},
        {"chunks, nt loads+stores", rw<0, 2, true>, 13312, 20480, 0, 2}, {"whole block, nt loads+stores", rw<1, 2, true>, 29120, 32768, 0, 2},
        {"chunks, out of place", rw<0, 0, false, true>, 13312, 20480, 0, 2}, {"chunks, out of place, nt", rw<0, 2, true, true>, 13312, 20480, 0, 2},
        {"whole block, out of place", rw<1, 0, false, true>, 29120, 32768, 0, 2}, {"whole block, out of place, nt", rw<1, 2, true, true>, 29120, 32768, 0, 2},
        {"chunks, read only", rw<0, 0, false, false, 1>, 13312, 20480, 0, 1}, {"chunks, read only nt", rw<0, 2, false, false, 1>, 13312, 20480, 0, 1},
        {"chunks, write only", rw<0, 0, false, false, 2>, 13312, 20480, 0, 1}, {"chunks, write only nt", rw<0, 0, true, false, 2>, 13312, 20480, 0, 1},
        {"chunks + 500 fma", rw<2>, 13312, 20480, 500, 2}, {"chunks + 250 fma", rw<2>, 13312, 20480, 250, 2},
    };
    std::printf("%lld blocks of %lld bytes (%.2f GB)\n%-32s chunk  lds/wave waves/CU spin   ms      TB/s\n", blocks, blk_bytes, blocks * blk_bytes / 1e9, "pattern");
    {   // aligned workgroups and register bursts
        struct Sp { const char* name; K k; int threads; long long grid; int lds; int chunk; };
        const long long total = blocks * blk_bytes;
        std::vector<Sp> sp = {
            {"4 waves aligned, chunks", rw_aligned<4, 0, false>, 256, blocks / 4, 4 * 13312, 13312},
            {"4 waves aligned, chunks, nt", rw_aligned<4, 2, true>, 256, blocks / 4, 4 * 13312, 13312},
            {"2 waves aligned, chunks, nt", rw_aligned<2, 2, true>, 128, blocks / 2, 2 * 13312, 13312},
            {"4 waves aligned, whole, nt", rw_aligned<4, 2, true>, 256, blocks / 4, 4 * 29120, 29120},
            {"burst 28 KB / workgroup", rw_burst<7, false>, 256, total / (7 * 4096), 0, 0},
            {"burst 28 KB / workgroup, nt", rw_burst<7, true>, 256, total / (7 * 4096), 0, 0},
            {"burst 112 KB / workgroup", rw_burst<28, false>, 256, total / (28 * 4096), 0, 0},
            {"burst 112 KB / workgroup, nt", rw_burst<28, true>, 256, total / (28 * 4096), 0, 0},
        };
        if (argc > 2 && argv[2][0] == 'n') {            // block_rw <bins> narrow: the narrow WPE kernel's pieces, block size 2244 B against 2304 B, plain against nt
            const long long bins = 1024LL * 513;
            std::printf("narrow WPE pattern: %lld bins, four per wavefront, 256-byte pieces\n", bins);
            for (int sb : {2244, 2304}) for (int nt = 0; nt < 2; ++nt) {
                const long long sbf = sb / 4;
                if (bins * sb > 2 * blocks * blk_bytes) { std::printf("buffer too small\n"); return 1; }
                auto launch = [&]() { if (nt) hipLaunchKernelGGL((rw_narrow<true>), dim3((unsigned)(bins / 16)), dim3(256), 0, 0, state, state2, sbf, 0, 0, 0.0f);
                                      else hipLaunchKernelGGL((rw_narrow<false>), dim3((unsigned)(bins / 16)), dim3(256), 0, 0, state, state2, sbf, 0, 0, 0.0f); };
                launch(); launch();
                CK(hipDeviceSynchronize());
                const int reps = 10;
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                const double moved = 2.0 * bins * (sb / 256) * 256.0;
                std::printf("block %4d B  %s   %7.3f ms   %6.3f TB/s of moved bytes   %8.1f k bins/ms\n", sb, nt ? "nt   " : "plain", ms, moved / (ms * 1e-3) / 1e12, bins / ms / 1e3);
            }
            return 0;
        }
        if (argc > 2 && argv[2][0] == 's') {            // block_rw <blocks> spin: only the burst-with-arithmetic sweep
            sp.clear();
            std::printf("burst of 28 KB per 256-thread workgroup, nt, SPIN dependent FMAs between loads and stores; lds bytes set the workgroups per CU\n");
            for (int lds : {4096, 20480, 40960}) for (int spin : {0, 250, 500, 1000, 2000, 4000}) {
                auto launch = [&]() { hipLaunchKernelGGL((rw_burst_spin<7, true>), dim3((unsigned)(total / (7 * 4096))), dim3(256), lds, 0, state, state2, blk_floats, 0, spin, 0.0f); };
                launch(); launch();
                CK(hipDeviceSynchronize());
                const int reps = 5;
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) launch();
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
                std::printf("lds %6d (<= %2d workgroups / CU)  spin %5d   %7.3f ms  %6.3f TB/s\n", lds, 163840 / lds > 8 ? 8 : 163840 / lds, spin, ms, 2.0 * blocks * blk_bytes / (ms * 1e-3) / 1e12);
            }
            return 0;
        }
        for (const Sp& c : sp) {
            if (c.lds > 65536) CK(hipFuncSetAttribute((const void*)c.k, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
            auto launch = [&]() { hipLaunchKernelGGL(c.k, dim3((unsigned)c.grid), dim3(c.threads), c.lds, 0, state, state2, blk_floats, c.chunk / 4, 0, 0.0f); };
            launch(); launch();
            CK(hipDeviceSynchronize());
            const int reps = 10;
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            std::printf("%-32s %6d %6d                   %7.3f  %6.3f\n", c.name, c.chunk, c.lds, ms, 2.0 * blocks * blk_bytes / (ms * 1e-3) / 1e12);
        }
    }
    for (const Case& c : cases) {
        auto launch = [&]() { hipLaunchKernelGGL(c.k, dim3((unsigned)blocks), dim3(64), c.lds_bytes, 0, state, state2, blk_floats, c.chunk_bytes / 4, c.spin, 0.0f); };
        launch(); launch();
        CK(hipDeviceSynchronize());
        const int reps = 10;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        int wpc = 163840 / c.lds_bytes; if (wpc > 32) wpc = 32;
        std::printf("%-32s %6d %6d   %3d      %5d  %7.3f  %6.3f\n", c.name, c.chunk_bytes, c.lds_bytes, wpc, c.spin, ms, c.traffic * blocks * blk_bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
