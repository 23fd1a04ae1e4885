// tools/ubench_hostpath.hip -- how should a caller-owned pageable HOST buffer reach the kernel?
// Times, for one buffer: (a) hipHostRegister cost, (b) the cycle kernel run directly on the
// registered host memory (zero-copy over PCIe, both directions at once), (c) hipMemcpy H2D + D2H of
// pageable memory, (d) the same from registered memory, (e) multi-threaded memcpy into pinned staging.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Imodulate_amd/csrc tools/ubench_hostpath.hip -o tools/ubench_hostpath -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "cycle_kernel_impl.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void launch_cycle(uint8_t *p, uint64_t n, hipStream_t st, uint32_t grid)
{
    CycleArgs a{};
    a.head_ptr = p; a.head_n = 0; a.body = p; a.body_words = n / 16; a.tail_ptr = p + n; a.tail_n = 0; a.lead = 0;
    a.base_head = a.base_body = a.base_tail = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)grid * 131072) % lcg::PERIOD);
    hipLaunchKernelGGL((modgpu_cycle_kernel<8, 1024, 2, true>), dim3(grid), dim3(1024), 0, st, a);
}

static void par_memcpy(void *d, const void *s, size_t n, int threads)
{
    std::vector<std::thread> th;
    size_t per = (n + threads - 1) / threads;
    per = (per + 4095) & ~size_t(4095);
    for (int t = 0; t < threads; ++t) {
        size_t o = t * per;
        if (o >= n) break;
        size_t l = std::min(per, n - o);
        th.emplace_back([=] { memcpy((char *)d + o, (const char *)s + o, l); });
    }
    for (auto &t : th) t.join();
}

int main(int argc, char **argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 30);
    hipStream_t st; CHECK(hipStreamCreate(&st));
    uint8_t *host = (uint8_t *)aligned_alloc(4096, n);
    for (uint64_t i = 0; i < n; i += 4096) host[i] = (uint8_t)i; // touch
    memset(host, 0x5a, n);
    uint8_t *dev; CHECK(hipMalloc(&dev, n));
    uint8_t *pinned; CHECK(hipHostMalloc((void **)&pinned, n, hipHostMallocDefault));
    memset(pinned, 1, n);
    double t0, t1;

    t0 = now(); CHECK(hipMemcpy(dev, host, n, hipMemcpyHostToDevice)); t1 = now();
    printf("hipMemcpy H2D pageable          %.4f s  %6.1f GB/s\n", t1 - t0, n / (t1 - t0) / 1e9);
    t0 = now(); CHECK(hipMemcpy(host, dev, n, hipMemcpyDeviceToHost)); t1 = now();
    printf("hipMemcpy D2H pageable          %.4f s  %6.1f GB/s\n", t1 - t0, n / (t1 - t0) / 1e9);
    t0 = now(); CHECK(hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice)); t1 = now();
    printf("hipMemcpy H2D pinned            %.4f s  %6.1f GB/s\n", t1 - t0, n / (t1 - t0) / 1e9);
    t0 = now(); CHECK(hipMemcpy(pinned, dev, n, hipMemcpyDeviceToHost)); t1 = now();
    printf("hipMemcpy D2H pinned            %.4f s  %6.1f GB/s\n", t1 - t0, n / (t1 - t0) / 1e9);
    for (int th : {1, 2, 4, 8, 16}) {
        t0 = now(); par_memcpy(pinned, host, n, th); t1 = now();
        printf("memcpy pageable->pinned x%-2d     %.4f s  %6.1f GB/s\n", th, t1 - t0, n / (t1 - t0) / 1e9);
    }
    for (uint64_t chunk : {n, (uint64_t)64 << 20}) {
        t0 = now();
        for (uint64_t o = 0; o < n; o += chunk) CHECK(hipHostRegister(host + o, std::min(chunk, n - o), hipHostRegisterDefault));
        t1 = now();
        printf("hipHostRegister chunks of %5llu MiB  %.4f s  %6.1f GB/s\n", (unsigned long long)(chunk >> 20), t1 - t0, n / (t1 - t0) / 1e9);
        if (chunk == n) {
            uint8_t *dp; CHECK(hipHostGetDevicePointer((void **)&dp, host, 0));
            for (uint32_t grid : {64u, 128u, 256u, 512u}) {
                launch_cycle(dp, n, st, grid); CHECK(hipStreamSynchronize(st));
                t0 = now(); launch_cycle(dp, n, st, grid); CHECK(hipStreamSynchronize(st)); t1 = now();
                printf("  zero-copy kernel on registered host mem grid=%3u  %.4f s  %6.1f GB/s payload\n", grid, t1 - t0, n / (t1 - t0) / 1e9);
            }
            t0 = now(); CHECK(hipMemcpy(dev, host, n, hipMemcpyHostToDevice)); t1 = now();
            printf("  hipMemcpy H2D registered      %.4f s  %6.1f GB/s\n", t1 - t0, n / (t1 - t0) / 1e9);
            t0 = now(); CHECK(hipMemcpy(host, dev, n, hipMemcpyDeviceToHost)); t1 = now();
            printf("  hipMemcpy D2H registered      %.4f s  %6.1f GB/s\n", t1 - t0, n / (t1 - t0) / 1e9);
        }
        t0 = now();
        for (uint64_t o = 0; o < n; o += chunk) CHECK(hipHostUnregister(host + o));
        t1 = now();
        printf("hipHostUnregister                %.4f s\n", t1 - t0);
    }
    // zero-copy on hipHostMalloc'd memory for comparison
    {
        uint8_t *dp; CHECK(hipHostGetDevicePointer((void **)&dp, pinned, 0));
        launch_cycle(dp, n, st, 256); CHECK(hipStreamSynchronize(st));
        t0 = now(); launch_cycle(dp, n, st, 256); CHECK(hipStreamSynchronize(st)); t1 = now();
        printf("zero-copy kernel on hipHostMalloc mem   %.4f s  %6.1f GB/s payload\n", t1 - t0, n / (t1 - t0) / 1e9);
    }
    return 0;
}
