// tools/tune_cycle.hip -- interleaved A/B timing of the cycle kernel's variants on one device:
// workgroup size, lane-words in flight, keystream instruction sequence, software pipelining,
// grid size, and the copy-only / compute-only ablations that bound it from the memory and the
// VALU side.  Instantiates the product's own device code (modulate_amd/csrc/cycle_kernel_impl.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Imodulate_amd/csrc tools/tune_cycle.hip -o tools/tune_cycle
// Run:   tools/tune_cycle [bytes=4294967296] [rounds=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cycle_kernel_impl.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

struct Variant {
    uint32_t base_off = 0;
    std::string name;
    void (*launch)(const CycleArgs &, uint32_t grid, hipStream_t);
    uint64_t chunk;
    uint32_t grid;
    std::vector<float> ms;
};

template <int U, int BLOCK, int ALG, int PIPE, int MODE, int SAUX = 16, int SYNC = 0> void launch(const CycleArgs &a, uint32_t grid, hipStream_t st)
{
    hipLaunchKernelGGL((modgpu_cycle_kernel<U, BLOCK, ALG, PIPE, MODE, SAUX, SYNC>), dim3(grid), dim3(BLOCK), 0, st, a);
}

int main(int argc, char **argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 32);
    int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const bool cold = argc > 3 && atoi(argv[3]) != 0; // evict the buffer from the Infinity Cache before every launch
    const int reps = cold ? 1 : (n >= (1ull << 30) ? 2 : 20);
    uint8_t *scratch = nullptr;
    if (cold) CHECK(hipMalloc(&scratch, 768ull << 20));
    uint8_t *buf;
    CHECK(hipMalloc(&buf, n + (1 << 20)));
    CHECK(hipMemset(buf, 0x5A, n));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    std::vector<Variant> vs;
#define ADD(U, B, ALG, PIPE, MODE, g)                                                                        \
    do {                                                                                                     \
        char b_[128];                                                                                        \
        snprintf(b_, sizeof b_, "%-7s U=%d B=%4d alg=%d pipe=%d grid=%5u", MODE == MODE_FULL ? "full" : MODE == MODE_COPY ? "copy" : "compute", U, B, ALG, (int)PIPE, (unsigned)(g)); \
        vs.push_back({0, b_, launch<U, B, ALG, PIPE, MODE>, (uint64_t)U * B * 16, (g), {}});                     \
    } while (0)
#define ADDS(U, B, ALG, PIPE, MODE, SAUX, g)                                                                  \
    do {                                                                                                     \
        char b_[128];                                                                                        \
        snprintf(b_, sizeof b_, "%-7s U=%d B=%4d alg=%d pipe=%d st=%2d grid=%5u", MODE == MODE_FULL ? "full" : MODE == MODE_COPY ? "copy" : "compute", U, B, ALG, (int)PIPE, SAUX, (unsigned)(g)); \
        vs.push_back({0, b_, launch<U, B, ALG, PIPE, MODE, SAUX>, (uint64_t)U * B * 16, (g), {}});               \
    } while (0)
#define ADDY(U, B, ALG, PIPE, MODE, SYNC, g)                                                                  \
    do {                                                                                                     \
        char b_[128];                                                                                        \
        snprintf(b_, sizeof b_, "%-7s U=%d B=%4d alg=%d pipe=%d sync=%d grid=%5u", MODE == MODE_FULL ? "full" : MODE == MODE_COPY ? "copy" : "compute", U, B, ALG, (int)PIPE, SYNC, (unsigned)(g)); \
        vs.push_back({0, b_, launch<U, B, ALG, PIPE, MODE, 16, SYNC>, (uint64_t)U * B * 16, (g), {}});           \
    } while (0)
    auto autogrid = [&](uint64_t chunk, uint32_t cap) { return (uint32_t)std::min<uint64_t>((n + chunk - 1) / chunk, cap); };
    ADDY(1, 256, 1, 0, MODE_FULL, 0, autogrid(4096, 16384));
    ADDY(2, 256, 1, 0, MODE_FULL, 0, autogrid(8192, 8192));
    ADDY(4, 256, 1, 0, MODE_FULL, 0, autogrid(16384, 4096));
    ADDY(1, 1024, 1, 0, MODE_FULL, 0, autogrid(16384, 4096));
    ADDY(2, 1024, 1, 0, MODE_FULL, 0, autogrid(32768, 2048));
    ADDY(4, 1024, 1, 0, MODE_FULL, 0, autogrid(65536, 1024));
    ADDY(4, 1024, 1, 0, MODE_FULL, 0, autogrid(65536, 512));
    ADDY(4, 1024, 1, 0, MODE_FULL, 0, autogrid(65536, 256));
    ADDY(4, 1024, 1, 2, MODE_FULL, 3, autogrid(65536, 256));
    ADDY(8, 1024, 1, 0, MODE_FULL, 0, autogrid(131072, 256));
    ADDY(8, 1024, 1, 2, MODE_FULL, 3, autogrid(131072, 256));
    CycleArgs a{};
    a.head_ptr = buf; a.head_n = 0; a.body = buf; a.body_words = n / 16; a.tail_ptr = buf + n; a.tail_n = 0; a.lead = 0;
    const uint32_t base0 = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    a.base_head = a.base_body = a.base_tail = base0;

    for (int r = 0; r < rounds + 1; ++r) {
        for (auto &v : vs) {
            a.body = buf + v.base_off;
            a.lead = (uint32_t)((uintptr_t)a.body & (v.chunk - 1));
            a.base_body = lcg::mulmod(base0, lcg::powmod(lcg::A, lcg::PERIOD - a.lead));
            a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)v.grid * v.chunk) % lcg::PERIOD);
            if (cold) CHECK(hipMemsetAsync(scratch, r, 768ull << 20, st));
            CHECK(hipEventRecord(e0, st));
            for (int k = 0; k < reps; ++k) v.launch(a, v.grid, st);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms / reps);
        }
    }
    CHECK(hipGetLastError());
    printf("bytes=%llu rounds=%d cold=%d  (GB/s = read+write = 2*bytes/t)\n", (unsigned long long)n, rounds, (int)cold);
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
        printf("%s  med %.4f ms  min %.4f ms  -> %7.1f GB/s (med) %7.1f (best)\n", v.name.c_str(), med, mn,
               2.0 * n / med / 1e6, 2.0 * n / mn / 1e6);
    }
    return 0;
}
