// tools/ubench_pcie_bidir.hip -- can the DMA engines run H2D and D2H at full rate at the same time?
// (the host-buffer path moves every byte across PCIe in both directions concurrently)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 1ull << 30, chunk = 16ull << 20;
    char *h0, *h1, *d0, *d1;
    CHECK(hipHostMalloc((void **)&h0, n, hipHostMallocDefault)); CHECK(hipHostMalloc((void **)&h1, n, hipHostMallocDefault));
    memset(h0, 1, n); memset(h1, 2, n);
    CHECK(hipMalloc((void **)&d0, n)); CHECK(hipMalloc((void **)&d1, n));
    hipStream_t s0, s1; CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            for (size_t o = 0; o < n; o += chunk) {
                if (mode != 1) CHECK(hipMemcpyAsync(d0 + o, h0 + o, chunk, hipMemcpyHostToDevice, s0));
                if (mode != 0) CHECK(hipMemcpyAsync(h1 + o, d1 + o, chunk, hipMemcpyDeviceToHost, s1));
            }
            CHECK(hipStreamSynchronize(s0)); CHECK(hipStreamSynchronize(s1));
            double dt = now() - t0;
            if (rep == 2) printf("%s: %.1f GB/s per direction (%.4f s per GiB)\n", mode == 0 ? "H2D only " : mode == 1 ? "D2H only " : "both ways", n / dt / 1e9, dt);
        }
    }
    return 0;
}
