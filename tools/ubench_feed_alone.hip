// tools/ubench_feed_alone.hip [MiB = 1024] [reps = 5] -- the host-fed kernel (the product's own TU, included) with NOBODY to wait for:
// every chunk marked ready before the launch, 8 "pipelines" x 2 page-locked slots of 256 KiB that the kernel cycles over and over, no
// host copies, no host threads.  What the kernel's trip loop itself carries across PCIe -- tickets, ready polls, fences, barriers and
// all -- beside the plain small-shape kernel in place on page-locked memory (tools/ubench_pcie_ceiling: 50.2-50.4 GB/s).  If this reads
// ~50 the host-fed routes' distance to the in-place ceiling is the host's copies; if it reads lower, it is the loop.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cycle_feed_kernel.hip"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const uint64_t n = (argc > 1 ? strtoull(argv[1], nullptr, 0) : 1024ull) << 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const uint32_t chunk = 256u << 10, pipes = 8;
    const uint64_t chunks = (n + chunk - 1) / chunk;
    if (chunks > 8192) { printf("at most 2 GiB\n"); return 1; }
    CycleFeedArgs a{};
    for (uint32_t k = 0; k < pipes * 2; ++k) {
        uint8_t *h;
        CHECK(hipHostMalloc((void **)&h, chunk, hipHostMallocPortable | hipHostMallocMapped));
        memset(h, (int)k, chunk);
        CHECK(hipHostGetDevicePointer((void **)&a.slot[k], h, 0));
    }
    uint32_t *flags, *flags_dev, *work;
    CHECK(hipHostMalloc((void **)&flags, (2 * 8192 + 16) * sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer((void **)&flags_dev, flags, 0));
    CHECK(hipMalloc((void **)&work, (8192 + 2) * sizeof(uint32_t)));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    a.ready = flags_dev;
    a.done = flags_dev + 8192;
    a.abort = flags_dev + 2 * 8192;
    a.work = work;
    a.n = n;
    a.patience_ticks = 100000000ull; // 1 s
    a.chunk_bytes = chunk;
    a.pipes = pipes;
    a.base = 12345;
    printf("== the host-fed kernel alone: %llu MiB through 16 page-locked slots of 256 KiB, every chunk ready at launch; GB/s of payload (= per direction), best / median of %d\n",
           (unsigned long long)(n >> 20), reps);
    for (uint32_t grid : {16u, 24u, 32u, 48u, 64u, 96u}) {
        std::vector<double> v;
        for (int r = 0; r < reps + 1; ++r) {
            for (uint64_t c = 0; c < chunks; ++c) flags[c] = 1, flags[8192 + c] = 0;
            flags[2 * 8192] = 0;
            CHECK(hipMemsetAsync(work, 0, (8192 + 2) * sizeof(uint32_t), st));
            CHECK(hipStreamSynchronize(st));
            const double t0 = now();
            CHECK(modgpu_launch_cycle_feed(a, grid, st));
            CHECK(hipStreamSynchronize(st));
            const double dt = now() - t0;
            uint64_t done = 0;
            for (uint64_t c = 0; c < chunks; ++c) done += flags[8192 + c];
            if (done != chunks) { printf("grid %u: only %llu of %llu chunks done\n", grid, (unsigned long long)done, (unsigned long long)chunks); return 1; }
            if (r) v.push_back(n / dt / 1e9);
        }
        std::sort(v.begin(), v.end());
        printf("   grid %3u   %6.2f  %6.2f\n", grid, v.back(), v[v.size() / 2]);
    }
    return 0;
}
