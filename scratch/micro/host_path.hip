// host_path.hip — what the host-buffer entry points (ds_process, ds_process_pcm16) can expect from this machine: H2D / D2H rates from pinned and
// pageable memory, the price of hipHostRegister / Unregister per call, and a host memcpy into pinned memory on 1 .. 8 threads.
//   hipcc -O2 --offload-arch=gfx950 host_path.hip -o host_path -lpthread && ./host_path
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main() {
    for (size_t mb : {1, 4, 16, 64}) {
        const size_t n = mb << 20;
        void *dev, *pin;
        CK(hipMalloc(&dev, n)); CK(hipHostMalloc(&pin, n, hipHostMallocDefault));
        char* pg = (char*)aligned_alloc(4096, n); memset(pg, 1, n); memset(pin, 1, n);
        hipStream_t s; CK(hipStreamCreate(&s));
        auto rate = [&](const char* what, auto f, int reps) {
            f(); CK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int i = 0; i < reps; ++i) f();
            CK(hipStreamSynchronize(s));
            const double dt = (now() - t0) / reps;
            printf("%3zu MB  %-46s %8.1f us  %6.2f GB/s\n", mb, what, dt * 1e6, n / dt / 1e9);
        };
        rate("H2D pinned (hipMemcpyAsync)", [&] { CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }, 20);
        rate("D2H pinned", [&] { CK(hipMemcpyAsync(pin, dev, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }, 20);
        rate("H2D pageable (hipMemcpyAsync)", [&] { CK(hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }, 10);
        rate("D2H pageable", [&] { CK(hipMemcpyAsync(pg, dev, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }, 10);
        rate("register + H2D + unregister", [&] { CK(hipHostRegister(pg, n, hipHostRegisterDefault)); CK(hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); CK(hipHostUnregister(pg)); }, 10);
        rate("hipHostRegister + Unregister alone", [&] { CK(hipHostRegister(pg, n, hipHostRegisterDefault)); CK(hipHostUnregister(pg)); }, 10);
        CK(hipHostRegister(pg, n, hipHostRegisterDefault));
        rate("H2D from a registered buffer", [&] { CK(hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); }, 20);
        CK(hipHostUnregister(pg));
        for (int th : {1, 2, 4, 8}) {
            char name[64]; snprintf(name, sizeof name, "host memcpy pageable -> pinned, %d thread(s)", th);
            rate(name, [&] { std::vector<std::thread> ts; for (int i = 0; i < th; ++i) ts.emplace_back([&, i] { memcpy((char*)pin + n / th * i, pg + n / th * i, n / th); }); for (auto& t : ts) t.join(); }, 10);
        }
        rate("H2D + D2H pinned, two streams at once (4:1)", [&] {
            static hipStream_t s2 = nullptr; if (!s2) CK(hipStreamCreate(&s2));
            CK(hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync((char*)pin + n / 2, (char*)dev + n / 2, n / 4, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); }, 20);
        free(pg); CK(hipHostFree(pin)); CK(hipFree(dev)); CK(hipStreamDestroy(s));
    }
    return 0;
}
