// tools/ubench_d2d.hip -- the node's device-to-device copy rate as the HIP runtime delivers it (SURVEY.md 8d asks
// for a measured copy ceiling beside the vendor peak): hipMemcpyDtoDAsync of N bytes, read + written bytes per second.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_d2d.hip -o tools/ubench_d2d      Run: tools/ubench_d2d [bytes=4294967296]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main(int argc, char **argv)
{
    size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 32);
    char *a, *b;
    CHECK(hipMalloc(&a, n));
    CHECK(hipMalloc(&b, n));
    CHECK(hipMemset(a, 1, n));
    CHECK(hipMemset(b, 2, n));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < 12; ++r) {
        CHECK(hipEventRecord(e0, st));
        CHECK(hipMemcpyDtoDAsync(b, a, n, st));
        CHECK(hipMemcpyDtoDAsync(a, b, n, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (r >= 2) ms.push_back(t / 2);
    }
    std::sort(ms.begin(), ms.end());
    printf("hipMemcpyDtoDAsync %zu bytes: median %.4f ms per copy -> %.1f GB/s read+write (best %.1f)\n", n, ms[ms.size() / 2],
           2.0 * n / ms[ms.size() / 2] / 1e6, 2.0 * n / ms.front() / 1e6);
    return 0;
}
