// tools/ubench_latency.hip -- what one kernel launch + wait costs on this node, whatever the kernel does: the floor under
// the per-call latency of modgpu_cycle_host on header-sized buffers.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_latency.hip -o tools/ubench_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1;} } while (0)
__global__ void k(unsigned* p, unsigned n) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = ~p[i]; }
__global__ void kflag(unsigned* p, unsigned n, volatile unsigned* flag, unsigned seq) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = ~p[i]; __threadfence_system(); if (i == 0) *flag = seq; }
template <typename F> double timeit(F f, int it = 2000) { for (int i = 0; i < 50; ++i) f(); auto t = std::chrono::steady_clock::now(); for (int i = 0; i < it; ++i) f(); return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count() / it; }
int main() {
    unsigned *pin; CHECK(hipHostMalloc((void**)&pin, 1 << 20, hipHostMallocPortable | hipHostMallocMapped));
    unsigned *flag; CHECK(hipHostMalloc((void**)&flag, 64, hipHostMallocPortable | hipHostMallocMapped)); *flag = 0;
    hipStream_t nb, bl; CHECK(hipStreamCreateWithFlags(&nb, hipStreamNonBlocking)); CHECK(hipStreamCreate(&bl));
    void* mapped;
    printf("hipHostGetDevicePointer          %.2f us\n", timeit([&] { (void)hipHostGetDevicePointer(&mapped, pin, 0); }));
    printf("launch+sync nonblocking stream   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, nb, pin, 1024u); (void)hipStreamSynchronize(nb); }));
    printf("launch+sync blocking stream      %.2f us\n", timeit([&] { hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, bl, pin, 1024u); (void)hipStreamSynchronize(bl); }));
    printf("launch+sync null stream          %.2f us\n", timeit([&] { hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, 0, pin, 1024u); (void)hipStreamSynchronize(0); }));
    unsigned seq = 0;
    printf("launch + spin on host flag       %.2f us\n", timeit([&] { ++seq; hipLaunchKernelGGL(kflag, dim3(4), dim3(256), 0, nb, pin, 1024u, flag, seq); while (*(volatile unsigned*)flag != seq) {} }));
    hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    printf("launch + event record+sync       %.2f us\n", timeit([&] { hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, nb, pin, 1024u); (void)hipEventRecord(ev, nb); (void)hipEventSynchronize(ev); }));
    printf("launch only (async, amortised)   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, nb, pin, 1024u); }, 200)); (void)hipStreamSynchronize(nb);
    return 0;
}
