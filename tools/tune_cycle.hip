// tools/tune_cycle.hip -- interleaved A/B timing of the cycle kernel's variants on one device:
// workgroup size, lane-words in flight, keystream instruction sequence, software pipelining,
// grid size, and the copy-only / compute-only ablations that bound it from the memory and the
// VALU side.  Instantiates the product's own device code (modulate_amd/csrc/cycle_kernel_impl.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Imodulate_amd/csrc tools/tune_cycle.hip -o tools/tune_cycle
// Run:   tools/tune_cycle [bytes=4294967296] [rounds=5]
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cycle_kernel_impl.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// Every variant is launched an even number of times, and the cipher is an involution: if a variant processed
// every chunk exactly once per launch the buffer is back to its fill pattern afterwards.  A variant whose
// schedule drops or repeats chunks (a racy hand-off, say) looks FAST precisely because it skips work, so its
// row is only printed as valid when this count is zero.
__global__ void count_mismatches(const uint32_t *p, uint64_t n_words, uint32_t want, unsigned long long *out)
{
    unsigned long long bad = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x) bad += p[i] != want;
    if (bad) atomicAdd(out, bad);
}

struct Variant {
    uint32_t base_off = 0;
    std::string name;
    void (*launch)(const CycleArgs &, uint32_t grid, hipStream_t);
    uint64_t chunk;
    uint32_t grid;
    std::vector<float> ms;
    unsigned long long bad = 0;
};

template <int U, int BLOCK, int ALG, int PIPE, int MODE, int SAUX = 16, int SYNC = 0, int TRACE = 0, int LDSW = 0> void launch(const CycleArgs &a, uint32_t grid, hipStream_t st)
{
    hipLaunchKernelGGL((modgpu_cycle_kernel<U, BLOCK, ALG, PIPE, MODE, SAUX, SYNC, TRACE, LDSW>), dim3(grid), dim3(BLOCK), 0, st, a);
}

// ALG: 2 = the shipped keystream sequence (canonicalising carry out of the fold, 3 instructions per byte), 1 = round 2's (4 per byte)
template <int U, int BLOCK, int TRACE = 0, int DEPTH = 1, int MODE = MODE_FULL, int SAUX = 16, int LAUX = 2, int B1 = 1, int B2 = 1, int ALG = 2> void launch_queue(const CycleArgs &a, uint32_t grid, hipStream_t st)
{
    // (the kernel takes a table of parts: one buffer planned as a CycleArgs is a table of one)
    hipLaunchKernelGGL((modgpu_cycle_queue_kernel<U, BLOCK, ALG, SAUX, TRACE, DEPTH, MODE, LAUX, B1, B2>), dim3(grid), dim3(BLOCK), 0, st,
                       cycle_queue_args_of(a, (uint32_t)U * BLOCK * 16));
}

// `tune_cycle trace <bytes> [grid]`: where a launch's time goes.  Runs the shipped streaming shape with
// TRACE=1 (lane 0 of each workgroup stamps wall_clock64 at start and after every trip's store burst) and
// prints, relative to the earliest start: dispatch skew, the first trip (fill), steady-state trips, the
// spread of finishing times (drain / imbalance), per XCD.
static int trace_main(uint64_t n, uint32_t grid_cap, bool queue)
{
    uint8_t *buf;
    uint64_t *trace, *h;
    CHECK(hipMalloc(&buf, n + (1 << 20)));
    CHECK(hipMemset(buf, 0x5A, n));
    const uint64_t chunk = (queue ? 4ull : 8ull) * 1024 * 16; // the shipped shapes: queue 64 KiB, static 128 KiB
    uint32_t grid = (uint32_t)std::min<uint64_t>((n + chunk - 1) / chunk, grid_cap);
    CHECK(hipMalloc(&trace, (size_t)grid * 32 * 8));
    h = (uint64_t *)malloc((size_t)grid * 32 * 8);
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CycleArgs a{};
    a.head_ptr = buf; a.body = buf; a.body_words = n / 16; a.tail_ptr = buf + n; a.trace = trace;
    CHECK(hipMalloc(&a.queue, 64));
    CHECK(hipMemset(a.queue, 0, 64));
    printf("== %s schedule\n", queue ? "work-queue" : "static");
    const uint32_t base0 = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    a.base_head = a.base_body = a.base_tail = base0;
    a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)grid * chunk) % lcg::PERIOD);
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemsetAsync(trace, 0, (size_t)grid * 32 * 8, st));
        // an untraced launch in front so the traced one runs back to back like in the bench (untraced: under the
        // queue a workgroup's trip count differs from launch to launch, stale stamps would survive)
        if (queue) launch_queue<4, 1024, 0, 1, MODE_FULL, 18>(a, grid, st); else launch<8, 1024, 1, 2, MODE_FULL, 16, 3, 0>(a, grid, st);
        CHECK(hipEventRecord(e0, st));
        if (queue) launch_queue<4, 1024, 1, 1, MODE_FULL, 18>(a, grid, st); else launch<8, 1024, 1, 2, MODE_FULL, 16, 3, 1>(a, grid, st);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, trace, (size_t)grid * 32 * 8, hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull;
        for (uint32_t b = 0; b < grid; ++b) t0 = std::min(t0, h[b * 32]);
        const uint64_t trips_max = (n / chunk + grid - 1) / grid;
        std::vector<double> start, first, last, dur;
        double xcd_end[8] = {}, xcd_n[8] = {};
        std::vector<std::vector<double>> trip_dur(trips_max + 1);
        for (uint32_t b = 0; b < grid; ++b) {
            const uint64_t *r = h + b * 32;
            uint32_t k = 1;
            while (k < 31 && r[k]) ++k; // stamps 1..k-1 are trip ends
            if (k < 2) continue;
            start.push_back((r[0] - t0) * 0.01);
            first.push_back((r[1] - r[0]) * 0.01);
            last.push_back((r[k - 1] - t0) * 0.01);
            for (uint32_t j = 2; j < k; ++j) trip_dur[std::min<uint64_t>(j, trips_max)].push_back((r[j] - r[j - 1]) * 0.01);
            xcd_end[r[31] & 7] += (r[k - 1] - t0) * 0.01;
            xcd_n[r[31] & 7] += 1;
        }
        auto stat = [](std::vector<double> v, const char *name) {
            if (v.empty()) return;
            std::sort(v.begin(), v.end());
            double sum = 0;
            for (double x : v) sum += x;
            printf("  %-22s min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f  mean %7.2f us  (n=%zu)\n", name, v.front(), v[v.size() / 10],
                   v[v.size() / 2], v[v.size() * 9 / 10], v.back(), sum / v.size(), v.size());
        };
        if (rep == 0) continue; // warm
        printf("bytes=%llu grid=%u trips/wg=%llu  launch (events) %.2f us -> %.1f GB/s r+w; traced stamps at 10 ns\n", (unsigned long long)n, grid,
               (unsigned long long)trips_max, ms * 1e3, 2.0 * n / ms / 1e6);
        stat(start, "wg start (skew)");
        stat(first, "first trip (fill)");
        for (uint64_t j = 2; j <= std::min<uint64_t>(trips_max, 4); ++j) {
            char nm[32];
            snprintf(nm, sizeof nm, "trip %llu%s", (unsigned long long)j, j == std::min<uint64_t>(trips_max, 4) && trips_max > 4 ? "+ (capped)" : "");
            stat(trip_dur[j], nm);
        }
        if (trips_max > 4) {
            std::vector<double> rest;
            for (uint64_t j = 5; j <= trips_max; ++j) rest.insert(rest.end(), trip_dur[j].begin(), trip_dur[j].end());
            stat(rest, "trips 5..last");
        }
        stat(last, "wg end");
        {
            std::vector<double> trips;
            for (uint32_t b = 0; b < grid; ++b) { const uint64_t *r = h + b * 32; uint32_t k = 1; while (k < 31 && r[k]) ++k; trips.push_back(k - 1); }
            stat(trips, "trips per wg (<=30 traced)");
        }
        printf("  mean wg end per XCD:");
        for (int x = 0; x < 8; ++x) printf(" %7.2f", xcd_n[x] ? xcd_end[x] / xcd_n[x] : 0.0);
        printf("\n");
    }
    return 0;
}

// `tune_cycle dvfs [bytes] [launches]`: how launch shapes ride through the clock transient that follows load onset
// (profiles/r03_first_pass.txt).  For each shape: 300 ms idle, then `launches` launches, each timed by its own pair
// of events, with a one-wave probe on a second stream sampling the shader clock (s_memtime ticks per 10 us of
// s_memrealtime).  Answers: is the dip's cost VALU-bound (copy-only rides through it, more workgroups help) or a
// property of the memory pipeline's clock?
__global__ void clock_probe(uint64_t *out, int samples, uint64_t period)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < samples; ++i) {
        const uint64_t t0 = wall_clock64(), c0 = clock64();
        uint64_t t1;
        do {
            __builtin_amdgcn_s_sleep(8);
            t1 = wall_clock64();
        } while (t1 - t0 < period);
        out[2 * i] = t1;
        out[2 * i + 1] = ((clock64() - c0) * 100) / (t1 - t0);
    }
}
__global__ void stamp_kernel(uint64_t *out) { if (threadIdx.x == 0) *out = wall_clock64(); }

static int dvfs_main(uint64_t n, int launches)
{
    uint8_t *buf;
    CHECK(hipMalloc(&buf, n + (1 << 20)));
    CHECK(hipMemset(buf, 0x5A, n));
    hipStream_t st, pst;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&pst, hipStreamNonBlocking));
    const int samples = 6000;
    uint64_t *d_probe, *d_stamp;
    CHECK(hipMalloc(&d_probe, samples * 16));
    CHECK(hipMalloc(&d_stamp, (launches + 1) * 8));
    CycleArgs a{};
    CHECK(hipMalloc(&a.queue, 64));
    CHECK(hipMemset(a.queue, 0, 64));
    a.head_ptr = buf; a.body = buf; a.body_words = n / 16; a.tail_ptr = buf + n;
    a.base_head = a.base_body = a.base_tail = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    struct Shape { const char *name; void (*launch)(const CycleArgs &, uint32_t, hipStream_t); uint32_t grid; uint32_t main = 0; uint32_t below = 0; };
    const Shape shapes[] = {
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers joining below 1850 MHz (shipped)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 256, 200, 1850},
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers joining below 1750 MHz", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 256, 200, 1750},
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers joining below 1900 MHz", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 256, 200, 1900},
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers joining below 1600 MHz", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 256, 200, 1600},
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers that never join", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 256, 200, 1},
        {"queue 64 KiB, FULL alg 2 (carry from the fold), grid 200, no helpers", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 200},
        {"queue 64 KiB, FULL alg 2, grid 208", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 208},
        {"queue 64 KiB, FULL alg 2, grid 216", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 216},
        {"queue 64 KiB, FULL alg 2, grid 224", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 224},
        {"queue 64 KiB, FULL alg 2, grid 256", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 256},
        {"queue 64 KiB, FULL alg 1 (round 2: shift + add), grid 200", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 1>, 200},
        {"queue 64 KiB, FULL alg 1, grid 224", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 1>, 224},
        {"queue 64 KiB, FULL alg 1, grid 256", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 1>, 256},
        {"queue 64 KiB, COPY-ONLY, grid 200", launch_queue<4, 1024, 0, 1, MODE_COPY, 18>, 200},
    };
    printf("bytes=%llu launches=%d: per launch  ms | GB/s (2*bytes/t) | shader MHz (mean of 10 us samples)\n", (unsigned long long)n, launches);
    for (const Shape &sh : shapes) {
        a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)sh.grid * 65536) % lcg::PERIOD);
        a.main_groups = sh.main;
        a.helper_below_mhz = sh.below;
        CHECK(hipDeviceSynchronize());
        usleep(300000);
        CHECK(hipMemsetAsync(d_probe, 0, samples * 16, pst));
        hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, pst, d_probe, samples, (uint64_t)1000);
        usleep(1000);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, st, d_stamp);
        for (int i = 0; i < launches; ++i) {
            sh.launch(a, sh.grid, st);
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, st, d_stamp + i + 1);
        }
        CHECK(hipStreamSynchronize(st));
        CHECK(hipStreamSynchronize(pst));
        std::vector<uint64_t> stamps(launches + 1), probe(samples * 2);
        CHECK(hipMemcpy(stamps.data(), d_stamp, stamps.size() * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(probe.data(), d_probe, probe.size() * 8, hipMemcpyDeviceToHost));
        printf("-- %s\n", sh.name);
        double total = 0;
        for (int i = 0; i < launches; ++i) {
            uint64_t sum = 0, cnt = 0;
            for (int k = 0; k < samples; ++k)
                if (probe[2 * k] > stamps[i] && probe[2 * k] <= stamps[i + 1]) { sum += probe[2 * k + 1]; ++cnt; }
            const double ms = (stamps[i + 1] - stamps[i]) / 1e5;
            total += ms;
            printf("   %2d  %.4f  %7.1f  %5llu\n", i + 1, ms, 2.0 * n / ms / 1e6, (unsigned long long)(cnt ? sum / cnt : 0));
        }
        printf("   all %d launches: %.4f ms = %.1f GB/s\n", launches, total, 2.0 * n * launches / total / 1e6);
        // an even number of launches must give the fill pattern back (a shape that drops or repeats chunks looks fast)
        unsigned long long *d_bad, bad = 0;
        CHECK(hipMalloc(&d_bad, 8));
        CHECK(hipMemset(d_bad, 0, 8));
        if (launches % 2) sh.launch(a, sh.grid, st);
        hipLaunchKernelGGL(count_mismatches, dim3(2048), dim3(256), 0, st, (const uint32_t *)buf, n / 4, 0x5A5A5A5Au, d_bad);
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
        CHECK(hipFree(d_bad));
        if (bad) printf("   ** INVALID: %llu words differ after an even number of passes **\n", bad);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "dvfs")
        return dvfs_main(argc > 2 ? strtoull(argv[2], nullptr, 0) : (1ull << 32), argc > 3 ? atoi(argv[3]) : 16);
    if (argc > 1 && std::string(argv[1]) == "trace")
        return trace_main(argc > 2 ? strtoull(argv[2], nullptr, 0) : (1ull << 32), argc > 3 ? (uint32_t)atoi(argv[3]) : 256u, argc > 4 && atoi(argv[4]) != 0);
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
    // the north_star's LDS write-combine stage, on the shipped shape: registers -> LDS -> registers -> store burst
    vs.push_back({0, "full    U=8 B=1024 alg=1 pipe=2 sync=3 +LDS stage grid=  256", launch<8, 1024, 1, 2, MODE_FULL, 16, 3, 0, 1>, 131072, autogrid(131072, 256), {}});
    vs.push_back({0, "full    U=4 B=1024 alg=1 pipe=2 sync=3 +LDS stage grid=  256", launch<4, 1024, 1, 2, MODE_FULL, 16, 3, 0, 1>, 65536, autogrid(65536, 256), {}});
// queue kernel variants: U, BLOCK, DEPTH, store policy, load policy, barrier before loads (B1), grid cap
#define ADDQX(U, B, D, SA, LA, B1, g)                                                                       \
    do {                                                                                                     \
        char b_[160];                                                                                        \
        snprintf(b_, sizeof b_, "queue   U=%2d B=%4d depth=%d st=%2d ld=%2d b1=%d %3d KiB chunks grid=%4u", U, B, D, SA, LA, B1, U * B * 16 / 1024, (unsigned)autogrid((uint64_t)U * B * 16, g)); \
        vs.push_back({0, b_, launch_queue<U, B, 0, D, MODE_FULL, SA, LA, B1>, (uint64_t)U * B * 16, autogrid((uint64_t)U * B * 16, g), {}}); \
    } while (0)
    ADDQX(4, 1024, 1, 16, 2, 1, 256); // shipped
    ADDQX(4, 1024, 1, 16, 2, 0, 256);
    ADDQX(8, 1024, 1, 16, 2, 1, 256);
    ADDQX(8, 512, 1, 16, 2, 1, 256);
    ADDQX(8, 512, 1, 16, 2, 0, 256);
    ADDQX(8, 512, 1, 16, 2, 1, 512);
    ADDQX(4, 1024, 2, 16, 2, 1, 256);
    ADDQX(4, 1024, 1, 18, 2, 1, 256);
    ADDQX(4, 1024, 1, 2, 2, 1, 256);
    ADDQX(4, 1024, 1, 0, 2, 1, 256);
    ADDQX(4, 1024, 1, 16, 0, 1, 256);
    ADDQX(4, 1024, 1, 16, 18, 1, 256);
    ADDQX(2, 1024, 1, 16, 2, 1, 256);
    ADDQX(4, 1024, 1, 18, 2, 1, 512); // 64 VGPRs: two workgroups fit a CU
    ADDQX(4, 1024, 1, 18, 2, 1, 240);
    ADDQX(4, 1024, 1, 18, 2, 1, 224);
    ADDQX(4, 1024, 1, 18, 2, 1, 208);
    ADDQX(4, 1024, 1, 18, 2, 1, 200);
    ADDQX(4, 1024, 1, 18, 2, 1, 192);
    ADDQX(4, 1024, 1, 18, 2, 1, 184);
    ADDQX(4, 1024, 1, 18, 2, 1, 176);
    ADDQX(4, 1024, 1, 18, 2, 1, 160);
    ADDQX(4, 1024, 1, 18, 2, 1, 144);
    ADDQX(4, 1024, 1, 18, 2, 1, 128);
    ADDQX(4, 1024, 2, 18, 2, 1, 192);
    ADDQX(4, 1024, 2, 18, 2, 1, 160);
    ADDQX(8, 1024, 1, 18, 2, 1, 192);
    ADDQX(8, 1024, 1, 18, 2, 1, 160);
    ADDQX(8, 512, 1, 18, 2, 1, 192);
    vs.push_back({0, "queue   U= 4 B=1024 st=18 ALG 2 (shipped: carry from the fold)  64 KiB grid= 200", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 65536, autogrid(65536, 200), {}});
    vs.push_back({0, "queue   U= 4 B=1024 st=18 ALG 1 (round 2: shift + add)          64 KiB grid= 200", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 1>, 65536, autogrid(65536, 200), {}});
    vs.push_back({0, "queue   U= 4 B=1024 st=18 ALG 2                                 64 KiB grid= 224", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 65536, autogrid(65536, 224), {}});
    vs.push_back({0, "queue   U= 4 B=1024 st=18 ALG 2                                 64 KiB grid= 256", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 65536, autogrid(65536, 256), {}});
    vs.push_back({0, "queue   U= 4 B=1024 st=18 b1=0, barrier BEHIND the store burst 64 KiB grid= 256", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 0, 2>, 65536, autogrid(65536, 256), {}});
    vs.push_back({0, "queue   U= 4 B=1024 st=18 b1=1, barrier BEHIND the store burst 64 KiB grid= 256", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 2>, 65536, autogrid(65536, 256), {}});
    vs.push_back({0, "queue   U= 8 B= 512 st=18 b1=0, barrier BEHIND the store burst 64 KiB grid= 256", launch_queue<8, 512, 0, 1, MODE_FULL, 18, 2, 0, 2>, 65536, autogrid(65536, 256), {}});
    vs.push_back({0, "queue   U= 4 B=1024 COPY-ONLY (no keystream) b1=1 64 KiB chunks grid= 256", launch_queue<4, 1024, 0, 1, MODE_COPY, 16, 2, 1>, 65536, autogrid(65536, 256), {}});
    vs.push_back({0, "queue   U= 4 B=1024 NO barriers: racy ticket hand-off    64 KiB grid= 256", launch_queue<4, 1024, 0, 1, MODE_FULL, 16, 2, 0, 0>, 65536, autogrid(65536, 256), {}});
    CycleArgs a{};
    CHECK(hipMalloc(&a.queue, 64));
    CHECK(hipMemset(a.queue, 0, 64));
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
    // validity pass: two launches of each variant on the freshly filled buffer must give the fill pattern back
    unsigned long long *d_bad;
    CHECK(hipMalloc(&d_bad, 8));
    for (auto &v : vs) {
        if (v.name.find("compute") == 0) continue; // the compute-only ablation never stores
        CHECK(hipMemsetAsync(buf, 0x5A, n, st));
        CHECK(hipMemsetAsync(d_bad, 0, 8, st));
        a.body = buf + v.base_off;
        a.lead = (uint32_t)((uintptr_t)a.body & (v.chunk - 1));
        a.base_body = lcg::mulmod(base0, lcg::powmod(lcg::A, lcg::PERIOD - a.lead));
        a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)v.grid * v.chunk) % lcg::PERIOD);
        v.launch(a, v.grid, st);
        v.launch(a, v.grid, st);
        hipLaunchKernelGGL(count_mismatches, dim3(2048), dim3(256), 0, st, (const uint32_t *)buf, n / 4, 0x5A5A5A5Au, d_bad);
        CHECK(hipMemcpyAsync(&v.bad, d_bad, 8, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
    }
    printf("bytes=%llu rounds=%d cold=%d  (GB/s = read+write = 2*bytes/t)\n", (unsigned long long)n, rounds, (int)cold);
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
        printf("%s  med %.4f ms  min %.4f ms  -> %7.1f GB/s (med) %7.1f (best)%s\n", v.name.c_str(), med, mn,
               2.0 * n / med / 1e6, 2.0 * n / mn / 1e6, v.bad ? "   ** INVALID: wrong results, timing meaningless **" : "");
    }
    return 0;
}
