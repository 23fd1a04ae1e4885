// first_pass.hip -- what the FIRST pass over a buffer costs, and why.
//
// bench.py's headline is a 40-launch steady state.  A real job (BASELINE configs 3-5) makes ONE pass per part,
// right after something else wrote the part.  This tool reproduces that regime through the product's C ABI
// (libmodgpu.so: modgpu_cycle_device) and separates the candidate causes:
//
//   series   fresh process: produce the buffer, then N back-to-back launches, each bracketed by timestamp
//            kernels (s_memrealtime, 100 MHz) so every launch has its own duration on the GPU's clock; a
//            one-wave probe kernel on a second stream samples the SHADER clock (s_memtime ticks per 10 us of
//            s_memrealtime) for the whole series, and a host thread samples sysfs (sclk, mclk, power) at ~1 kHz.
//            -> clock / power ramp, or not.
//   first    per size and per producer (hipMemcpy H2D from pinned memory, hipMemset, a fill kernel with plain /
//            nt / sc1+nt stores, the cycle kernel itself), optionally with an idle gap or an Infinity-Cache
//            scrub between producer and consumer: ONE launch, timed.  -> dirty-line write-back, or not.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/first_pass.hip -o tools/first_pass -Lmodulate_amd -lmodgpu -Wl,-rpath,'$ORIGIN/../modulate_amd' -lpthread
// Run:   tools/first_pass series [bytes=4294967296] [launches=16] [producer=h2d]
//        tools/first_pass first
#include <hip/hip_runtime.h>

#include <dirent.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "modgpu.h"

#define CHECK(x)                                                                                                      \
    do {                                                                                                              \
        hipError_t e_ = (x);                                                                                          \
        if (e_ != hipSuccess) {                                                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));                         \
            exit(1);                                                                                                  \
        }                                                                                                             \
    } while (0)
#define MOD(x)                                                                                                        \
    do {                                                                                                              \
        int r_ = (x);                                                                                                 \
        if (r_ != 0) {                                                                                                \
            fprintf(stderr, "%s:%d %s: modgpu error %d (%s)\n", __FILE__, __LINE__, #x, r_, modgpu_last_error());    \
            exit(1);                                                                                                  \
        }                                                                                                             \
    } while (0)

static const int32_t KEY = (int32_t)0x90CFC0ABu;

__global__ void stamp_kernel(uint64_t *out) { if (threadIdx.x == 0) *out = wall_clock64(); }

// one wave: shader-clock ticks per `period` ticks of the 100 MHz real-time counter, `samples` times (bounded: it ends by itself)
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
        const uint64_t c1 = clock64();
        out[2 * i] = t1;
        out[2 * i + 1] = ((c1 - c0) * 100) / (t1 - t0); // MHz
    }
}

using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
// fill with a pattern; AUX = cache-policy bits of the stores (0 plain, 2 nt, 16 sc1, 18 sc1+nt)
template <int AUX> __global__ __launch_bounds__(1024) void fill_kernel(uint8_t *p, uint64_t n16, uint32_t seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        u32x4 v{(uint32_t)i * 2654435761u + seed, (uint32_t)(i >> 7) ^ seed, seed, (uint32_t)i};
        auto r = __builtin_amdgcn_make_buffer_rsrc(p + ((i * 16) & ~0x3FFFFFFFull), 0, 0x40000000, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)((i * 16) & 0x3FFFFFFFull), 0, AUX);
    }
}
// reads `n16` words (scrub: pushes whatever the Infinity Cache held out of it)
__global__ __launch_bounds__(1024) void read_kernel(const u32x4 *p, uint64_t n16, uint32_t *sink)
{
    u32x4 acc{0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) acc ^= p[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

static std::string find_hwmon()
{
    for (int card = 0; card < 16; ++card) {
        std::string base = "/sys/class/drm/card" + std::to_string(card) + "/device/hwmon";
        DIR *d = opendir(base.c_str());
        if (!d) continue;
        std::string found;
        while (dirent *e = readdir(d))
            if (strncmp(e->d_name, "hwmon", 5) == 0) found = base + "/" + e->d_name;
        closedir(d);
        if (!found.empty() && access((found + "/freq1_input").c_str(), R_OK) == 0) return found;
    }
    return "";
}
static long read_long(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    long v = -1;
    if (fscanf(f, "%ld", &v) != 1) v = -1;
    fclose(f);
    return v;
}

enum Producer { P_H2D, P_MEMSET, P_FILL_PLAIN, P_FILL_NT, P_FILL_SC1NT, P_CYCLE, P_COUNT };
static const char *kProducerName[P_COUNT] = {"hipMemcpy H2D (pinned)", "hipMemset", "fill kernel, plain stores", "fill kernel, nt stores",
                                             "fill kernel, sc1+nt stores", "cycle kernel (a previous pass)"};
static Producer parse_producer(const char *s)
{
    const char *names[P_COUNT] = {"h2d", "memset", "fill", "fill_nt", "fill_sc1nt", "cycle"};
    for (int i = 0; i < P_COUNT; ++i)
        if (strcmp(s, names[i]) == 0) return (Producer)i;
    fprintf(stderr, "producer: h2d | memset | fill | fill_nt | fill_sc1nt | cycle\n");
    exit(2);
}

struct Ctx {
    uint8_t *buf = nullptr;
    uint64_t cap = 0;
    uint8_t *pinned = nullptr; // 64 MiB of host pattern
    uint8_t *scrub = nullptr;  // 1 GiB of other memory
    uint32_t *sink = nullptr;
    hipStream_t st = nullptr;
};
static const uint64_t kTile = 64ull << 20;

static void produce(Ctx &c, Producer p, uint64_t n)
{
    switch (p) {
    case P_H2D:
        for (uint64_t off = 0; off < n; off += kTile) CHECK(hipMemcpyAsync(c.buf + off, c.pinned, std::min(kTile, n - off), hipMemcpyHostToDevice, c.st));
        break;
    case P_MEMSET: CHECK(hipMemsetAsync(c.buf, 0x5A, n, c.st)); break;
    case P_FILL_PLAIN: hipLaunchKernelGGL(fill_kernel<0>, dim3(1024), dim3(1024), 0, c.st, c.buf, n / 16, 7u); break;
    case P_FILL_NT: hipLaunchKernelGGL(fill_kernel<2>, dim3(1024), dim3(1024), 0, c.st, c.buf, n / 16, 7u); break;
    case P_FILL_SC1NT: hipLaunchKernelGGL(fill_kernel<18>, dim3(1024), dim3(1024), 0, c.st, c.buf, n / 16, 7u); break;
    case P_CYCLE: MOD(modgpu_cycle_device(c.buf, n, KEY, 0, -1, c.st)); break;
    default: break;
    }
    CHECK(hipGetLastError());
}

// ONE launch over [buf, buf+n), timed by events on the launch stream
static float one_launch_ms(Ctx &c, uint64_t n)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, c.st));
    MOD(modgpu_cycle_device(c.buf, n, KEY, 0, -1, c.st));
    CHECK(hipEventRecord(e1, c.st));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms;
}

static int first_main(Ctx &c)
{
    const uint64_t sizes[] = {256ull << 20, 411ull * 1000 * 1000, 512ull << 20, 1ull << 30, 1ull << 32};
    enum Between { B_NONE, B_IDLE, B_SCRUB, B_COUNT };
    const char *between_name[B_COUNT] = {"at once", "after 20 ms idle", "after a 1 GiB scrub read"};
    // warm the chip (clocks, page tables of the whole buffer) so that what differs between rows is the producer
    for (int i = 0; i < 12; ++i) MOD(modgpu_cycle_device(c.buf, c.cap, KEY, 0, -1, c.st));
    CHECK(hipStreamSynchronize(c.st));
    printf("== first pass after a producer: ONE modgpu_cycle_device launch, HIP events on the launch stream; GB/s = 2*bytes/time\n");
    printf("   (steady = the same launch repeated: mean of launches 3..8 of 8 back-to-back)\n");
    for (uint64_t n : sizes) {
        // steady state at this size
        for (int i = 0; i < 2; ++i) MOD(modgpu_cycle_device(c.buf, n, KEY, 0, -1, c.st));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, c.st));
        for (int i = 0; i < 6; ++i) MOD(modgpu_cycle_device(c.buf, n, KEY, 0, -1, c.st));
        CHECK(hipEventRecord(e1, c.st));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("-- %6.0f MiB   steady %8.1f GB/s (%.4f ms)\n", n / 1048576.0, 2.0 * n / (ms / 6 * 1e-3) / 1e9, ms / 6);
        for (int p = 0; p < P_COUNT; ++p)
            for (int b = 0; b < B_COUNT; ++b) {
                float best = 1e9f, worst = 0, sum = 0;
                const int reps = 5;
                for (int r = 0; r < reps; ++r) {
                    produce(c, (Producer)p, n);
                    CHECK(hipStreamSynchronize(c.st));
                    if (b == B_IDLE) usleep(20000);
                    if (b == B_SCRUB) {
                        hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(1024), 0, c.st, reinterpret_cast<const u32x4 *>(c.scrub), (1ull << 30) / 16, c.sink);
                        CHECK(hipStreamSynchronize(c.st));
                    }
                    const float t = one_launch_ms(c, n);
                    best = std::min(best, t);
                    worst = std::max(worst, t);
                    sum += t;
                }
                printf("   %-32s %-26s mean %8.1f GB/s   best %8.1f   worst %8.1f\n", kProducerName[p], between_name[b],
                       2.0 * n / (sum / reps * 1e-3) / 1e9, 2.0 * n / (best * 1e-3) / 1e9, 2.0 * n / (worst * 1e-3) / 1e9);
            }
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
    return 0;
}

static int series_main(Ctx &c, uint64_t n, int launches, Producer prod, int pre_idle_ms)
{
    hipStream_t probe_st;
    CHECK(hipStreamCreateWithFlags(&probe_st, hipStreamNonBlocking));
    const int samples = 6000; // x 10 us = 60 ms: covers the producer's tail and the whole series
    uint64_t *d_probe, *d_stamps;
    CHECK(hipMalloc(&d_probe, samples * 16));
    CHECK(hipMemset(d_probe, 0, samples * 16));
    CHECK(hipMalloc(&d_stamps, (launches + 1) * 8));
    // host sampler: sysfs clocks and power
    const std::string hw = find_hwmon();
    struct Smp { double t_ms; long sclk, mclk, power; };
    std::vector<Smp> smp;
    std::atomic<bool> stop{false};
    const auto t_origin = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_origin).count(); };
    std::thread sampler([&] {
        if (hw.empty()) return;
        while (!stop.load()) {
            Smp s{now_ms(), read_long(hw + "/freq1_input"), read_long(hw + "/freq2_input"), read_long(hw + "/power1_average")};
            if (s.power < 0) s.power = read_long(hw + "/power1_input");
            smp.push_back(s);
            usleep(300);
        }
    });
    produce(c, prod, n);
    CHECK(hipStreamSynchronize(c.st));
    if (pre_idle_ms > 0) usleep(pre_idle_ms * 1000);
    const double t_series0 = now_ms();
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, probe_st, d_probe, samples, (uint64_t)1000);
    usleep(2000); // the probe shows 2 ms of idle clock before the first launch
    const double t_launch0 = now_ms();
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, c.st, d_stamps);
    for (int i = 0; i < launches; ++i) {
        MOD(modgpu_cycle_device(c.buf, n, KEY, 0, -1, c.st));
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, c.st, d_stamps + i + 1);
    }
    CHECK(hipStreamSynchronize(c.st));
    const double t_done = now_ms();
    CHECK(hipStreamSynchronize(probe_st));
    stop.store(true);
    sampler.join();
    std::vector<uint64_t> stamps(launches + 1), probe(samples * 2);
    CHECK(hipMemcpy(stamps.data(), d_stamps, stamps.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(probe.data(), d_probe, probe.size() * 8, hipMemcpyDeviceToHost));
    printf("== series: producer = %s, %d launches of %.0f MiB, idle before the series %d ms\n", kProducerName[prod], launches, n / 1048576.0, pre_idle_ms);
    printf("   launch  start(us)  duration(ms)   GB/s    shader clock during it (MHz: min / mean / max of 10 us samples)\n");
    for (int i = 0; i < launches; ++i) {
        const uint64_t a = stamps[i], b = stamps[i + 1];
        uint64_t lo = ~0ull, hi = 0, sum = 0, cnt = 0;
        for (int k = 0; k < samples; ++k)
            if (probe[2 * k] > a && probe[2 * k] <= b) {
                lo = std::min(lo, probe[2 * k + 1]);
                hi = std::max(hi, probe[2 * k + 1]);
                sum += probe[2 * k + 1];
                ++cnt;
            }
        const double ms = (b - a) / 1e5;
        printf("   %5d  %9.1f  %10.4f  %8.1f   %5llu / %5llu / %5llu  (%llu samples)\n", i + 1, (a - stamps[0]) / 100.0, ms, 2.0 * n / (ms * 1e-3) / 1e9,
               (unsigned long long)(cnt ? lo : 0), (unsigned long long)(cnt ? sum / cnt : 0), (unsigned long long)(cnt ? hi : 0), (unsigned long long)cnt);
    }
    // shader clock in the 2 ms before the first launch
    {
        uint64_t sum = 0, cnt = 0;
        for (int k = 0; k < samples; ++k)
            if (probe[2 * k] && probe[2 * k] < stamps[0]) {
                sum += probe[2 * k + 1];
                ++cnt;
            }
        printf("   shader clock while idle, just before launch 1: %llu MHz (%llu samples)\n", (unsigned long long)(cnt ? sum / cnt : 0), (unsigned long long)cnt);
    }
    printf("   host view: series launched at %.2f ms, done at %.2f ms (probe started %.2f)\n", t_launch0, t_done, t_series0);
    if (hw.empty()) printf("   sysfs hwmon not readable here: no mclk / power samples\n");
    else {
        printf("   sysfs (%s): t(ms rel. to launch 1)  sclk(MHz)  mclk(MHz)  power(W)   [%zu samples, ~%.0f Hz]\n", hw.c_str(), smp.size(),
               smp.size() > 1 ? (smp.size() - 1) / ((smp.back().t_ms - smp.front().t_ms) * 1e-3) : 0.0);
        double last_print = -1e9;
        long prev_s = -2, prev_m = -2;
        for (const Smp &s : smp) {
            const double rel = s.t_ms - t_launch0;
            if (rel < -5 || rel > (t_done - t_launch0) + 5) continue;
            const bool changed = s.sclk != prev_s || s.mclk != prev_m;
            if (changed || rel - last_print >= 1.0) {
                printf("      %8.2f  %8ld  %8ld  %8.1f\n", rel, s.sclk / 1000000, s.mclk / 1000000, s.power / 1e6);
                last_print = rel;
            }
            prev_s = s.sclk;
            prev_m = s.mclk;
        }
    }
    CHECK(hipFree(d_probe));
    CHECK(hipFree(d_stamps));
    CHECK(hipStreamDestroy(probe_st));
    return 0;
}

// ---- wake: what the chip's FIRST launch after an upload pays, and whether a one-wave kernel enqueued by the producer takes it away
// (BENCH_r04: 411 MB 0.1941 ms as the first launch after the upload, 0.1259 ms for the launch right after).
// Every trial: `idle_ms` of nothing (the shader engines go to sleep), then the part is uploaded the way bench.py does it
// (64 MiB tiles, synchronous hipMemcpy -- what modgpu_h2d was until round 5), then ONE modgpu_cycle_device launch timed by HIP events on its stream AND by two stamp
// kernels' wall clocks.  Variants, interleaved:
//   plain        nothing else                                       (what round 4's library does)
//   at_start     a one-wave kernel on another stream when the upload STARTS (async; the engines have the whole upload to wake)
//   each_tile    the same before every tile (an upload of seconds could let them fall asleep again)
//   at_end       a one-wave kernel right after the last tile, not waited for
//   waited       a one-wave kernel after the last tile, WAITED for, then the launch: the wake-up paid outside the timed launch --
//                the floor any trick can reach; its own duration is the wake-up itself
__global__ void nop_kernel(uint32_t *sink) { if (sink && threadIdx.x == 12345u) *sink = 1; }
// one word per 64 KiB of a buffer: the address translations of its pages are walked before the first real launch needs them
__global__ void touch_kernel(const uint32_t *p, uint64_t blocks, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < blocks; i += (uint64_t)gridDim.x * blockDim.x) acc ^= p[i * 16384];
    if (acc == 0x9E3779B9u) *sink = acc;
}
static int wake_main(Ctx &c, uint64_t n, int trials, int idle_ms)
{
    hipStream_t side;
    CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    const char *names[] = {"plain", "at_start", "each_tile", "at_end", "waited", "library", "fresh", "fresh_touch", "fresh_2nd", "null_stream"};
    const int V = 10; // "null_stream": as "library", the launch on the legacy NULL stream (where bench.py's first_pass launches) // "library": the upload through modgpu_h2d, which (since round 5) does each_tile itself; the others upload with hipMemcpy.
    // "fresh": as "library", into a buffer allocated for this trial (hipMalloc) -- what bench.py's part is: memory no shader has touched
    // yet, so the first launch also walks cold page tables; "fresh_touch": the same, and a side-stream kernel reads one word per 64 KiB
    // of it after the upload (not waited for); "fresh_2nd": fresh, and the number reported is the SECOND launch's (for comparison)
    std::vector<std::vector<float>> ms(V), nop_us(V), next_ms(V);
    auto nop = [&](hipStream_t st) { hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, st, c.sink); };
    // WAKE_ONLY=<variant index>: that variant alone (with trials = 1: the PROCESS's first large launch, what bench.py's first_pass times)
    const int only = getenv("WAKE_ONLY") ? atoi(getenv("WAKE_ONLY")) : -1;
    if (only >= 0) { // the part as bench.py has it: from modgpu_alloc, which also prepares the device (code object, ring, both kernels resolved)
        void *p = nullptr;
        const uint64_t want = getenv("WAKE_UPLOAD_BYTES") ? std::max<uint64_t>(n, strtoull(getenv("WAKE_UPLOAD_BYTES"), nullptr, 0)) : n;
        MOD(modgpu_alloc(&p, want, 0));
        c.buf = (uint8_t *)p;
        c.cap = want;
        if (getenv("WAKE_WARM")) { // a REAL work-queue launch (two 64 KiB parts share one: ring line, sign-off word and all) before anything is timed
            void *parts[2] = {c.buf, c.buf + (1 << 20)};
            uint64_t sizes[2] = {65536, 65536};
            for (int k = 0; k < 2; ++k) MOD(modgpu_cycle_batch_device(parts, sizes, nullptr, 2, KEY, 0, c.st));
            CHECK(hipStreamSynchronize(c.st));
        }
    }
    for (int t = 0; t < trials; ++t)
        for (int v = 0; v < V; ++v) {
            if (only >= 0 && v != only) {
                ms[v].push_back(0), next_ms[v].push_back(0), nop_us[v].push_back(0);
                continue;
            }
            usleep(idle_ms * 1000);
            uint8_t *const keep = c.buf;
            if (v >= 6 && v <= 8) CHECK(hipMalloc(&c.buf, n));
            hipStream_t const keep_st = c.st;
            if (v == 9) c.st = nullptr;
            if (v == 1) nop(side);
            // WAKE_PAGEABLE=1: the upload's source is ordinary memory (bench.py's numpy tile); WAKE_UPLOAD_BYTES=B: the upload covers B
            // bytes of the buffer although only the first n are cycled (bench.py uploads its whole 4 GiB part, then cycles 411 MB of it)
            static uint8_t *const pageable = getenv("WAKE_PAGEABLE") ? (uint8_t *)memset(malloc(kTile), 0x37, kTile) : nullptr;
            const uint64_t up = getenv("WAKE_UPLOAD_BYTES") ? std::min<uint64_t>(strtoull(getenv("WAKE_UPLOAD_BYTES"), nullptr, 0), c.cap) : n;
            const uint8_t *from = pageable ? pageable : c.pinned;
            for (uint64_t off = 0; off < up; off += kTile) {
                if (v == 2) nop(side);
                if (v >= 5) MOD(modgpu_h2d(c.buf + off, from, std::min(kTile, up - off), 0));
                else CHECK(hipMemcpy(c.buf + off, from, std::min(kTile, up - off), hipMemcpyHostToDevice));
            }
            if (v == 7) hipLaunchKernelGGL(touch_kernel, dim3(64), dim3(256), 0, side, (const uint32_t *)c.buf, n / 65536, c.sink);
            float w = 0;
            if (v == 3) nop(side);
            if (v == 4) {
                const auto t0 = std::chrono::steady_clock::now();
                nop(side);
                CHECK(hipStreamSynchronize(side));
                w = std::chrono::duration<float, std::micro>(std::chrono::steady_clock::now() - t0).count();
            }
            const float first = one_launch_ms(c, n), second = one_launch_ms(c, n); // the launch right after undoes the first: the size's own rate
            ms[v].push_back(v == 8 ? second : first);
            next_ms[v].push_back(second);
            nop_us[v].push_back(w);
            CHECK(hipStreamSynchronize(side));
            c.st = keep_st;
            if (v >= 6 && v <= 8) {
                CHECK(hipFree(c.buf));
                c.buf = keep;
            }
        }
    printf("== wake: %d trials per variant, interleaved; %d ms idle, upload of %.0f MiB in 64 MiB tiles, then ONE launch (HIP events)\n", trials, idle_ms, n / 1048576.0);
    printf("   %-10s  %28s  %28s  %s\n", "variant", "first launch: median / min / max ms", "next launch: median ms", "GB/s (median first)   frac of 8 TB/s");
    for (int v = 0; v < V; ++v) {
        auto med = [](std::vector<float> x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
        const float m = med(ms[v]);
        printf("   %-10s  %10.4f / %7.4f / %7.4f  %28.4f  %10.1f  %.4f", names[v], m, *std::min_element(ms[v].begin(), ms[v].end()), *std::max_element(ms[v].begin(), ms[v].end()),
               med(next_ms[v]), 2.0 * n / (m * 1e-3) / 1e9, 2.0 * n / (m * 1e-3) / 8e12);
        if (v == 4) printf("   (the waited-for one-wave kernel: median %.1f us by the host's clock)", med(nop_us[v]));
        printf("\n");
    }
    CHECK(hipStreamDestroy(side));
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: first_pass series [bytes] [launches] [producer] [idle_ms] | first_pass first | first_pass wake [bytes] [trials] [idle_ms]\n");
        return 2;
    }
    const bool series = strcmp(argv[1], "series") == 0, wake = strcmp(argv[1], "wake") == 0;
    Ctx c;
    c.cap = (series || wake) && argc > 2 ? strtoull(argv[2], nullptr, 0) : 1ull << 32;
    CHECK(hipSetDevice(0));
    CHECK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
    CHECK(hipMalloc(&c.buf, c.cap));
    CHECK(hipMalloc(&c.scrub, 1ull << 30));
    CHECK(hipMalloc(&c.sink, 4));
    CHECK(hipMemset(c.scrub, 1, 1ull << 30));
    CHECK(hipHostMalloc(&c.pinned, kTile, hipHostMallocDefault));
    for (uint64_t i = 0; i < kTile; ++i) c.pinned[i] = (uint8_t)(i * 131 + 7);
    CHECK(hipDeviceSynchronize());
    int rc;
    if (wake) rc = wake_main(c, c.cap, argc > 3 ? atoi(argv[3]) : 7, argc > 4 ? atoi(argv[4]) : 1500);
    else if (series) rc = series_main(c, c.cap, argc > 3 ? atoi(argv[3]) : 16, argc > 4 ? parse_producer(argv[4]) : P_H2D, argc > 5 ? atoi(argv[5]) : 0);
    else rc = first_main(c);
    modgpu_path_stats_t st;
    MOD(modgpu_path_stats(&st, 0));
    printf("   engine: %llu kernel launches through libmodgpu.so, %llu host-loop calls\n", (unsigned long long)st.gpu_launches, (unsigned long long)st.scalar_calls);
    return rc;
}
