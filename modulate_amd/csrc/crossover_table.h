// crossover_table.h -- what MODGPU_HOST_POLICY=fastest decides by: the two engines' measured rates for a caller-owned HOST
// buffer on the node class this library is tuned on (MI355X behind PCIe gen 5 x16, 2 x EPYC 9575F), from
// profiles/r05_small_call_crossover.txt (kernel route) and r04_small_call_crossover.txt (host loop) -- bin/modbench --hostcall; re-measure
// there and edit here for another host.
// Payload GB/s, one call, warm.  Not used by the default policy (offload), which only has the size threshold.
#pragma once
#include <cstdint>

namespace crossover {

// kernel route, PAGEABLE caller memory (staged: copy -> pinned slot -> kernel across PCIe on the slot -> copy back)
struct Point { uint64_t bytes; double gbps; };
constexpr Point kKernelPageable[] = { // end of round 5 (one host-fed kernel per call below 2 GiB, a launch per 8 MiB chunk above): profiles/r05_small_call_crossover.txt
    {4ull << 20, 29.2}, {8ull << 20, 36.7}, {16ull << 20, 41.5}, {32ull << 20, 45.0}, {64ull << 20, 46.9}, {128ull << 20, 47.8}, {256ull << 20, 48.4}, {1024ull << 20, 48.6}, {4096ull << 20, 49.3},
};
// kernel route, PAGE-LOCKED caller memory (modgpu_host_alloc / _register): one kernel across PCIe where the pages lie
constexpr double kKernelPinnedGbps = 50.0;     // profiles/r02_sweep_pinned_routes.txt
constexpr double kKernelCallOverheadUs = 14.0; // launch + wait before a byte moves (profiles/r02_ubench_latency.txt)
// host loop, ONE thread, by body (generic, avx2, avx512), and what its threads reach together before DRAM is the bound
constexpr double kHostThreadGbps[3] = {1.6, 8.5, 17.4};
constexpr double kHostThreadsEfficiency = 0.6; // of threads x one thread's rate: 8 threads 113 GB/s at 16 MiB (0.81), 16 threads 127 / 169 at 32 / 64 MiB (0.46 / 0.61)
constexpr double kHostDramGbps = 170.0;        // 16 threads, 64 MiB ... 1 GiB: 145-180 GB/s
constexpr double kHostWakeUs = 25.0;           // waking parked workers and waiting for the last of them

inline double kernel_pageable_gbps(uint64_t n)
{
    constexpr int N = (int)(sizeof kKernelPageable / sizeof kKernelPageable[0]);
    if (n <= kKernelPageable[0].bytes) return kKernelPageable[0].gbps * (double)n / (double)kKernelPageable[0].bytes; // latency-bound below the table
    for (int i = 1; i < N; ++i)
        if (n <= kKernelPageable[i].bytes) {
            const double f = (double)(n - kKernelPageable[i - 1].bytes) / (double)(kKernelPageable[i].bytes - kKernelPageable[i - 1].bytes);
            return kKernelPageable[i - 1].gbps + f * (kKernelPageable[i].gbps - kKernelPageable[i - 1].gbps);
        }
    return kKernelPageable[N - 1].gbps;
}
// microseconds one call over n bytes takes on each engine
inline double kernel_us(uint64_t n, bool pinned)
{
    return pinned ? kKernelCallOverheadUs + (double)n / (kKernelPinnedGbps * 1e3) : (double)n / (kernel_pageable_gbps(n) * 1e3);
}
inline double host_us(uint64_t n, int isa, unsigned threads)
{
    const double one = kHostThreadGbps[isa < 0 || isa > 2 ? 0 : isa];
    if (threads <= 1) return (double)n / (one * 1e3);
    double all = (double)threads * one * kHostThreadsEfficiency;
    if (all > kHostDramGbps) all = kHostDramGbps;
    return kHostWakeUs + (double)n / (all * 1e3);
}

} // namespace crossover
