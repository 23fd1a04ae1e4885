// tools/tune_cycle.hip -- interleaved A/B timing of the cycle kernel's variants on one device:
// workgroup size, lane-words in flight, keystream instruction sequence, software pipelining,
// grid size, and the copy-only / compute-only ablations that bound it from the memory and the
// VALU side.  Instantiates the product's own kernels (modulate_amd/csrc/cycle_kernel_impl.h: the "PRODUCT" rows) and the
// lab forms of them with their tuning knobs (tools/cycle_kernel_lab.h).
// Build: make -C tools tune_cycle
// Run:   tools/tune_cycle [bytes=4294967296] [rounds=5] [cold=0]
//        tools/tune_cycle trace <bytes> [grid=256] [queue=0] [variant=0] [tail_chunks=0] [cold=0] [dump=0]
//        tools/tune_cycle dvfs [bytes] [launches]
//        tools/tune_cycle selftest          (the validity checks must reject a deliberately wrong keystream)
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "cycle_kernel_lab.h"
#include "golden_kat.inc" // KAT_PS4_FNV_1M, KAT_PS4_FNV_AT_2G_1M: from tests/golden/cycle_golden.json (tools/Makefile)

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
    void (*launch)(const LabArgs &, uint32_t grid, hipStream_t);
    uint64_t chunk;
    uint32_t grid;
    std::vector<float> ms;
    unsigned long long bad = 0;    // words that differ from the fill pattern after two passes (involution)
    uint32_t tail = 0;             // LabArgs::tail_chunks (TSPLIT variants)
    unsigned long long ks_bad = 0; // bytes of one pass over zeros that are not the reference keystream
};

template <int U, int BLOCK, int ALG, int PIPE, int MODE, int SAUX = 16, int SYNC = 0, int TRACE = 0, int LDSW = 0> void launch(const LabArgs &a, uint32_t grid, hipStream_t st)
{
    hipLaunchKernelGGL((lab_cycle_kernel<U, BLOCK, ALG, PIPE, MODE, SAUX, SYNC, TRACE, LDSW>), dim3(grid), dim3(BLOCK), 0, st, a);
}

// ALG: 2 = the shipped keystream sequence (canonicalising carry out of the fold, 3 instructions per byte), 1 = round 2's (4 per byte)
template <int U, int BLOCK, int TRACE = 0, int DEPTH = 1, int MODE = MODE_FULL, int SAUX = 16, int LAUX = 2, int B1 = 1, int B2 = 1, int ALG = 2, int TSPLIT = 0, int TK = 1, int TLOOP = 0, int LSP = 0, int HSB = 0>
void launch_queue(const LabArgs &a, uint32_t grid, hipStream_t st)
{
    // (the kernel takes a table of parts: one buffer planned as a CycleArgs is a table of one)
    hipLaunchKernelGGL((lab_cycle_queue_kernel<U, BLOCK, ALG, SAUX, TRACE, DEPTH, MODE, LAUX, B1, B2, TSPLIT, TK, TLOOP, LSP, HSB>), dim3(grid), dim3(BLOCK), 0, st,
                       lab_queue_args_of(a, (uint32_t)U * BLOCK * 16));
}
// the kernels the product ships, instantiated from the product header itself
void launch_product_queue(const LabArgs &a, uint32_t grid, hipStream_t st)
{
    hipLaunchKernelGGL((modgpu_cycle_queue_kernel<4, 1024>), dim3(grid), dim3(1024), 0, st, lab_queue_table_of(a, 65536));
}
void launch_product_large(const LabArgs &a, uint32_t grid, hipStream_t st)
{
    hipLaunchKernelGGL((modgpu_cycle_kernel<8, 1024, 2, true>), dim3(grid), dim3(1024), 0, st, static_cast<const CycleArgs &>(a));
}
void launch_product_small(const LabArgs &a, uint32_t grid, hipStream_t st)
{
    hipLaunchKernelGGL((modgpu_cycle_kernel<1, 256, 1, false>), dim3(grid), dim3(256), 0, st, static_cast<const CycleArgs &>(a));
}

// ---- validity of a variant's OUTPUT.  The involution check above cannot see a wrong keystream (a variant
// that XORs the same wrong bytes twice restores the pattern).  So every variant is also run ONCE over zero bytes and its
// output -- the keystream itself -- is compared, over the whole buffer, with a reference keystream computed on the fly by
// a deliberately plain kernel (64-bit `%`, square-and-multiply: none of the product's folds, tables or SDWA packing), and
// that reference is itself pinned to the reference implementation's known answers from tests/golden (FNV-1a-64 of the
// PS4 keystream's first MiB, and of the MiB at 2^31 when the buffer reaches it).  No code under oracle/ is involved.
__device__ uint32_t ref_mulmod(uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)x * y) % 0x7FFFFFFFull); }
__device__ uint32_t ref_powmod(uint32_t b, uint64_t e)
{
    uint32_t r = 1;
    for (; e; e >>= 1) {
        if (e & 1) r = ref_mulmod(r, b);
        b = ref_mulmod(b, b);
    }
    return r;
}
// one thread per 64 bytes: counts bytes of `got` that differ from the keystream of `key_res` at stream positions pos0 + i;
// write != nullptr: stores the reference keystream there instead of comparing
__global__ void ref_keystream_check(const uint8_t *got, uint8_t *write, uint64_t n, uint32_t key_res, uint64_t pos0, unsigned long long *bad)
{
    unsigned long long mine = 0;
    for (uint64_t blk = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; blk * 64 < n; blk += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i0 = blk * 64;
        uint32_t s = ref_mulmod(ref_powmod(16807u, ((pos0 + i0) % 0x7FFFFFFEull) + 1), key_res);
        for (uint64_t i = i0; i < i0 + 64 && i < n; ++i) {
            const uint8_t ks = (uint8_t)~s;
            if (write) write[i] = ks;
            else mine += got[i] != ks;
            s = ref_mulmod(s, 16807u);
        }
    }
    if (mine) atomicAdd(bad, mine);
}
static uint64_t fnv1a64(const uint8_t *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}
// the reference kernel against the golden digests; exits on a mismatch (nothing below could be trusted)
static void pin_reference_kernel(uint8_t *scratch, uint64_t n, hipStream_t st)
{
    const uint32_t key_res = lcg::key_residue((int32_t)0x90cfc0ab);
    std::vector<uint8_t> h(1 << 20);
    struct { uint64_t pos, want; } kats[2] = {{0, KAT_PS4_FNV_1M}, {1ull << 31, KAT_PS4_FNV_AT_2G_1M}};
    for (auto &k : kats) {
        hipLaunchKernelGGL(ref_keystream_check, dim3(256), dim3(64), 0, st, (const uint8_t *)nullptr, scratch, (uint64_t)(1 << 20), key_res, k.pos, (unsigned long long *)nullptr);
        CHECK(hipMemcpyAsync(h.data(), scratch, 1 << 20, hipMemcpyDeviceToHost, st));
        CHECK(hipStreamSynchronize(st));
        const uint64_t got = fnv1a64(h.data(), h.size());
        if (got != k.want) {
            printf("reference keystream kernel disagrees with tests/golden at stream position %llu: %016llx, want %016llx\n", (unsigned long long)k.pos,
                   (unsigned long long)got, (unsigned long long)k.want);
            exit(2);
        }
    }
    (void)n;
}
// bytes of buf[0..n) that are not the PS4 keystream from stream position 0 (buf was zero, one launch of the variant ran)
static unsigned long long keystream_mismatches(const uint8_t *buf, uint64_t n, unsigned long long *d_bad, hipStream_t st)
{
    unsigned long long bad = 0;
    CHECK(hipMemsetAsync(d_bad, 0, 8, st));
    hipLaunchKernelGGL(ref_keystream_check, dim3(4096), dim3(256), 0, st, buf, (uint8_t *)nullptr, n, lcg::key_residue((int32_t)0x90cfc0ab), (uint64_t)0, d_bad);
    CHECK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
    CHECK(hipStreamSynchronize(st));
    return bad;
}

// `tune_cycle trace <bytes> [grid]`: where a launch's time goes.  Runs the shipped streaming shape with
// TRACE=1 (lane 0 of each workgroup stamps wall_clock64 at start and after every trip's store burst) and
// prints, relative to the earliest start: dispatch skew, the first trip (fill), steady-state trips, the
// spread of finishing times (drain / imbalance), per XCD.
// variant (queue schedule): 0 = the product's loop, 1 = TSPLIT 1 (halves over the last tail_chunks chunks), 2 = TSPLIT 2 (quarters),
// 3 = TK (ticket fetched at the start of the trip), 4 = TK + TSPLIT 1, 5 / 6 = TLOOP with halves / quarters (pieces in a second, cold loop).  cold: a 768 MB memset evicts the buffer from the
// Infinity Cache before every traced launch (and nothing runs back to back in front of it).  dump: one line per workgroup.
static int trace_main(uint64_t n, uint32_t grid_cap, bool queue, int variant, uint32_t tail_chunks, bool cold, bool dump)
{
    uint8_t *buf;
    uint64_t *trace, *h;
    CHECK(hipMalloc(&buf, n + (1 << 20)));
    CHECK(hipMemset(buf, 0x5A, n));
    const uint64_t chunk = (queue ? 4ull : 8ull) * 1024 * 16; // the shipped shapes: queue 64 KiB, static 128 KiB
    uint32_t grid = (uint32_t)std::min<uint64_t>((n + chunk - 1) / chunk, grid_cap);
    CHECK(hipMalloc(&trace, (size_t)grid * kTraceSlots * 8));
    h = (uint64_t *)malloc((size_t)grid * kTraceSlots * 8);
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    LabArgs a{};
    a.head_ptr = buf; a.body = buf; a.body_words = n / 16; a.tail_ptr = buf + n; a.trace = trace;
    CHECK(hipMalloc(&a.queue, 64));
    CHECK(hipMemset(a.queue, 0, 64));
    a.tail_chunks = tail_chunks;
    uint8_t *scrub = nullptr;
    if (cold) CHECK(hipMalloc(&scrub, 768ull << 20));
    using LaunchFn = void (*)(const LabArgs &, uint32_t, hipStream_t);
    static const struct { const char *name; LaunchFn plain, traced; } kQueueVariants[] = {
        {"the product's loop", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18>},
        {"TSPLIT 1: tail chunks handed out as halves", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 0>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 0>},
        {"TSPLIT 2: tail chunks handed out as quarters", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 2, 0>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18, 2, 1, 1, 2, 2, 0>},
        {"TK: ticket fetched at the start of the trip", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1>},
        {"TK + TSPLIT 1", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 1>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 1>},
        {"TLOOP + TSPLIT 1: halves in a second, cold loop", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 0, 1>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 0, 1>},
        {"TLOOP + TSPLIT 2: quarters in a second, cold loop", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 2, 0, 1>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18, 2, 1, 1, 2, 2, 0, 1>},
        {"HSB 2 diagnostic: 56 helpers join after sleeping 30 % of the launch (grid 256 = 200 main + 56)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 2>, launch_queue<4, 1024, 1, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 2>},
    };
    if (variant < 0 || variant > 7) variant = 0;
    if (variant == 7) {
        a.main_groups = 200;
        a.helper_below_mhz = 1850;
        a.standby_ticks = (uint32_t)((double)n / 3.5e12 * 1e8 * 0.30);
    }
    printf("== %s schedule%s%s, tail_chunks=%u, %s\n", queue ? "work-queue" : "static", queue ? ": " : "", queue ? kQueueVariants[variant].name : "", tail_chunks,
           cold ? "COLD (768 MB memset in front of each traced launch)" : "warm, back to back behind an untraced launch");
    const uint32_t base0 = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    a.base_head = a.base_body = a.base_tail = base0;
    a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)grid * chunk) % lcg::PERIOD);
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipMemsetAsync(trace, 0, (size_t)grid * kTraceSlots * 8, st));
        // an untraced launch in front so the traced one runs back to back like in the bench (untraced: under the
        // queue a workgroup's trip count differs from launch to launch, stale stamps would survive)
        if (cold) CHECK(hipMemsetAsync(scrub, rep, 768ull << 20, st));
        else if (queue) kQueueVariants[variant].plain(a, grid, st);
        else launch<8, 1024, 2, 2, MODE_FULL, 16, 3, 0>(a, grid, st);
        CHECK(hipEventRecord(e0, st));
        if (queue) kQueueVariants[variant].traced(a, grid, st); else launch<8, 1024, 2, 2, MODE_FULL, 16, 3, 1>(a, grid, st);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, trace, (size_t)grid * kTraceSlots * 8, hipMemcpyDeviceToHost));
        uint64_t t0 = ~0ull;
        for (uint32_t b = 0; b < grid; ++b) t0 = std::min(t0, h[b * kTraceSlots]);
        const uint64_t trips_max = (n / chunk + grid - 1) / grid;
        std::vector<double> start, first, last, dur;
        double xcd_end[8] = {}, xcd_n[8] = {};
        std::vector<std::vector<double>> trip_dur(trips_max + 1);
        for (uint32_t b = 0; b < grid; ++b) {
            const uint64_t *r = h + b * kTraceSlots;
            uint32_t k = 1;
            while (k < kTraceSlots - 1 && r[k]) ++k; // stamps 1..k-1 are trip ends
            if (k < 2) continue;
            start.push_back((r[0] - t0) * 0.01);
            first.push_back((r[1] - r[0]) * 0.01);
            last.push_back((r[k - 1] - t0) * 0.01);
            for (uint32_t j = 2; j < k; ++j) trip_dur[std::min<uint64_t>(j, trips_max)].push_back((r[j] - r[j - 1]) * 0.01);
            xcd_end[r[kTraceSlots - 1] & 7] += (r[k - 1] - t0) * 0.01;
            xcd_n[r[kTraceSlots - 1] & 7] += 1;
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
            for (uint32_t b = 0; b < grid; ++b) { const uint64_t *r = h + b * kTraceSlots; uint32_t k = 1; while (k < kTraceSlots - 1 && r[k]) ++k; trips.push_back(k - 1); }
            stat(trips, "trips per wg (<=62 traced)");
        }
        printf("  mean wg end per XCD:");
        for (int x = 0; x < 8; ++x) printf(" %7.2f", xcd_n[x] ? xcd_end[x] / xcd_n[x] : 0.0);
        printf("\n");
        {   // where the launch's time goes, from the stamps: before the first workgroup starts and after the last one ends the
            // events see only launch overhead; between the mean end and the last end the chip drains
            double first_start = 1e30, last_end = 0, mean_end = 0, mean_first = 0;
            for (size_t i = 0; i < last.size(); ++i) { first_start = std::min(first_start, start[i]); last_end = std::max(last_end, last[i]); mean_end += last[i]; mean_first += first[i]; }
            mean_end /= last.size();
            mean_first /= last.size();
            const double ideal = 2.0 * n / 7.0e6; // us at 7.0 TB/s, the 4 GiB steady state
            printf("  split: events %.2f us = first wg start .. last wg end %.2f us + %.2f us outside the stamps (launch, signal);"
                   " mean first trip %.2f us; mean wg end %.2f, last %.2f: finishing spread %.2f us; ideal at 7.0 TB/s %.2f us\n",
                   ms * 1e3, last_end - first_start, ms * 1e3 - (last_end - first_start), mean_first, mean_end, last_end, last_end - mean_end, ideal);
        }
        if (dump && rep == 3) {
            printf("  per workgroup: blk xcc start_us first_trip_end_us trips end_us last_three_trip_durations_us\n");
            for (uint32_t b = 0; b < grid; ++b) {
                const uint64_t *r = h + b * kTraceSlots;
                uint32_t k = 1;
                while (k < kTraceSlots - 1 && r[k]) ++k;
                if (k < 2) { printf("   %3u %llu  (no trip)\n", b, (unsigned long long)(r[kTraceSlots - 1] & 7)); continue; }
                printf("   %3u %llu %6.2f %6.2f %2u %7.2f ", b, (unsigned long long)(r[kTraceSlots - 1] & 7), (r[0] - t0) * 0.01, (r[1] - t0) * 0.01, k - 1, (r[k - 1] - t0) * 0.01);
                for (uint32_t j = (k > 4 ? k - 3 : 2); j < k; ++j) printf(" %5.2f", (r[j] - r[j - 1]) * 0.01);
                printf("\n");
            }
        }
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
    LabArgs a{};
    CHECK(hipMalloc(&a.queue, 64));
    CHECK(hipMemset(a.queue, 0, 64));
    a.head_ptr = buf; a.body = buf; a.body_words = n / 16; a.tail_ptr = buf + n;
    a.base_head = a.base_body = a.base_tail = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    struct Shape { const char *name; void (*launch)(const LabArgs &, uint32_t, hipStream_t); uint32_t grid; uint32_t main = 0; uint32_t below = 0; uint32_t standby_pct = 0; };
    const Shape shapes[] = {
        // round 5: helpers that STAND BY -- asleep, no memory traffic -- and look at the clock again every ~50 us for standby_pct % of the launch's
        // expected duration (n / 3.5 TB/s), instead of looking once; "never join" rows price the standing by itself.  Grid 248 (31 workgroups per XCD), not 256:
        // the clock probe's wave occupies a CU of one XCD for the whole series, so that XCD's 32nd persistent 1024-thread workgroup would not start
        // before another one there has left -- harmless when helpers decide at once, but a helper that stands by would BEGIN its stand-by when the
        // launch is already over and stretch it by that long (what the first runs of this experiment measured: profiles/r05_standby.txt)
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers below 1850 MHz, decide once (shipped)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 248, 200, 1850},
        {"... standing by for 80 % of the launch", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1850, 80},
        {"... standing by for 50 % of the launch", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1850, 50},
        {"... standing by for 40 % of the launch", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1850, 40},
        {"... standing by for 30 % of the launch", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1850, 30},
        {"... standing by for 40 % of the launch, never joining (what standing by costs)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1, 40},
        {"... standing by for 20 % of the launch, never joining (what standing by costs)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1, 20},
        {"... standing by for 80 % of the launch, never joining (what standing by costs)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1, 80},
        {"... decide once, never joining", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 248, 200, 1},
        // diagnostic (HSB 2): every helper joins, whatever the clock, after sleeping for pct % of the launch -- what a LATE joiner does to a launch
        {"... diag: all 56 helpers join at once (0 %)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 2>, 248, 200, 1850, 0},
        {"... diag: all 56 helpers join after 10 % of the launch", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 2>, 248, 200, 1850, 10},
        {"... diag: all 56 helpers join after 30 % of the launch", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 2>, 248, 200, 1850, 30},
        {"... diag: all 56 helpers join after 60 % of the launch", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 2>, 248, 200, 1850, 60},
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers below 1850 MHz, decide once (shipped), again", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 248, 200, 1850},
        {"... standing by for 80 % of the launch, again", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 0, 1>, 248, 200, 1850, 80},
        {"queue 64 KiB, FULL alg 2, 200 main + 56 helpers joining below 1850 MHz (shipped)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 256, 200, 1850},
        {"the same with round 3's ticket timing (TK 0)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 0>, 256, 200, 1850},
        {"PRODUCT kernel, 200 main + 56 helpers below 1850 MHz", launch_product_queue, 256, 200, 1850},
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
        a.standby_ticks = (uint32_t)((double)n / 3.5e12 * 1e8 * sh.standby_pct / 100.0);
        if (getenv("DVFS_ONLY_STANDBY") && !(sh.standby_pct || strstr(sh.name, "decide once") || strstr(sh.name, "diag"))) continue;
        if (getenv("DVFS_ONLY_DIAG") && !(strstr(sh.name, "diag") || strstr(sh.name, "(shipped)"))) continue;
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

// `tune_cycle selftest`: the validity checks must be able to fail.  A kernel that XORs a keystream shifted by one byte -- an
// involution like the real one, so check (1) passes it -- must be caught by check (2), and the real product kernels must pass.
__global__ void wrong_keystream_kernel(uint8_t *buf, uint64_t n, uint32_t key_res)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        buf[i] ^= (uint8_t)~ref_mulmod(ref_powmod(16807u, ((i + 1) % 0x7FFFFFFEull) + 1), key_res); // position i+1 instead of i
}
static int selftest_main()
{
    const uint64_t n = 64ull << 20;
    uint8_t *buf;
    unsigned long long *d_bad, bad = 0;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    CHECK(hipMalloc(&buf, n + (1 << 20)));
    CHECK(hipMalloc(&d_bad, 8));
    pin_reference_kernel(buf + n, n, st);
    const uint32_t key_res = lcg::key_residue((int32_t)0x90cfc0ab);
    int failures = 0;
    // the deliberately wrong kernel: passes the involution check, must fail the keystream check
    CHECK(hipMemsetAsync(buf, 0x5A, n, st));
    CHECK(hipMemsetAsync(d_bad, 0, 8, st));
    hipLaunchKernelGGL(wrong_keystream_kernel, dim3(2048), dim3(256), 0, st, buf, n, key_res);
    hipLaunchKernelGGL(wrong_keystream_kernel, dim3(2048), dim3(256), 0, st, buf, n, key_res);
    hipLaunchKernelGGL(count_mismatches, dim3(2048), dim3(256), 0, st, (const uint32_t *)buf, n / 4, 0x5A5A5A5Au, d_bad);
    CHECK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
    CHECK(hipStreamSynchronize(st));
    printf("wrong-keystream kernel: involution check sees %llu bad words (expected 0: it cannot see this class of error)\n", bad);
    failures += bad != 0;
    CHECK(hipMemsetAsync(buf, 0, n, st));
    hipLaunchKernelGGL(wrong_keystream_kernel, dim3(2048), dim3(256), 0, st, buf, n, key_res);
    bad = keystream_mismatches(buf, n, d_bad, st);
    printf("wrong-keystream kernel: keystream check sees %llu bad bytes of %llu (expected: nearly all) -> %s\n", bad, (unsigned long long)n, bad > n / 2 ? "REJECTED, as it must be" : "** NOT CAUGHT **");
    failures += !(bad > n / 2);
    // the product kernels must pass it
    LabArgs a{};
    CHECK(hipMalloc(&a.queue, 64));
    CHECK(hipMemset(a.queue, 0, 64));
    a.head_ptr = buf; a.body = buf; a.body_words = n / 16; a.tail_ptr = buf + n;
    a.base_head = a.base_body = a.base_tail = lcg::state_residue(key_res, 0);
    struct { const char *name; void (*fn)(const LabArgs &, uint32_t, hipStream_t); uint64_t chunk; uint32_t grid; } prod[] = {
        {"modgpu_cycle_queue_kernel<4, 1024>", launch_product_queue, 65536, 200},
        {"modgpu_cycle_kernel<8, 1024, 2, true>", launch_product_large, 131072, 256},
        {"modgpu_cycle_kernel<1, 256, 1, false>", launch_product_small, 4096, 16384}};
    for (auto &k : prod) {
        CHECK(hipMemsetAsync(buf, 0, n, st));
        a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)k.grid * k.chunk) % lcg::PERIOD);
        k.fn(a, k.grid, st);
        bad = keystream_mismatches(buf, n, d_bad, st);
        printf("%s: %llu bad bytes -> %s\n", k.name, bad, bad ? "** FAILS **" : "ok");
        failures += bad != 0;
    }
    printf(failures ? "SELFTEST FAILED\n" : "SELFTEST OK\n");
    return failures ? 1 : 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "dvfs")
        return dvfs_main(argc > 2 ? strtoull(argv[2], nullptr, 0) : (1ull << 32), argc > 3 ? atoi(argv[3]) : 16);
    if (argc > 1 && std::string(argv[1]) == "trace")
        return trace_main(argc > 2 ? strtoull(argv[2], nullptr, 0) : (1ull << 32), argc > 3 ? (uint32_t)atoi(argv[3]) : 256u, argc > 4 && atoi(argv[4]) != 0,
                          argc > 5 ? atoi(argv[5]) : 0, argc > 6 ? (uint32_t)atoi(argv[6]) : 0u, argc > 7 && atoi(argv[7]) != 0, argc > 8 && atoi(argv[8]) != 0);
    if (argc > 1 && std::string(argv[1]) == "selftest") return selftest_main();
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
    // ---- what the product ships, from the product header itself, beside the lab form of the same loop
    vs.push_back({0, "PRODUCT modgpu_cycle_queue_kernel<4, 1024>                     64 KiB grid= 200", launch_product_queue, 65536, autogrid(65536, 200), {}});
    vs.push_back({0, "lab form of it (every knob at the product's setting)           64 KiB grid= 200", launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 65536, autogrid(65536, 200), {}});
    vs.push_back({0, "PRODUCT modgpu_cycle_kernel<8, 1024, 2, true> (static map)     128 KiB grid= 256", launch_product_large, 131072, autogrid(131072, 256), {}});
    vs.push_back({0, "PRODUCT modgpu_cycle_kernel<1, 256, 1, false> (small shape)      4 KiB grid=16384", launch_product_small, 4096, autogrid(4096, 16384), {}});
    // ---- round 4, the tail of a launch: finer pieces where the work runs out, and one chunk less committed
    {
        const uint32_t g = autogrid(65536, 200);
        auto add_tail = [&](const char *name, void (*fn)(const LabArgs &, uint32_t, hipStream_t), uint32_t tail) {
            char b_[160];
            snprintf(b_, sizeof b_, "queue   %-44s tail=%4u chunks grid= %u", name, tail, g);
            Variant v{0, b_, fn, 65536, g, {}};
            v.tail = tail;
            vs.push_back(v);
        };
        for (uint32_t t : {g / 2, g, 2 * g, 3 * g, 4 * g}) add_tail("TSPLIT 1 (halves at the tail)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 0>, t);
        for (uint32_t t : {g / 4, g / 2, g, 2 * g}) add_tail("TSPLIT 2 (quarters at the tail)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 2, 0>, t);
        add_tail("TK 1 (ticket at the start of the trip: product)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1>, 0);
        add_tail("TK 0 (round 3: ticket published a trip later)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 0>, 0);
        for (uint32_t t : {g, 2 * g, 3 * g}) add_tail("TK + TSPLIT 1", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 1>, t);
        add_tail("TK + TSPLIT 2", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 2, 1>, g / 2);
        add_tail("TSPLIT 1 over the WHOLE buffer (32 KiB pieces)", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 0>, 0xFFFFFFFFu);
        // the pieces in a second, cold loop behind the trip loop (the trip loop itself as the product's)
        for (uint32_t t : {g / 4, g / 2, g, 3 * g / 2, 2 * g, 3 * g}) add_tail("TLOOP halves in a cold second loop", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 1, 0, 1>, t);
        for (uint32_t t : {g / 4, g / 2, g, 2 * g}) add_tail("TLOOP quarters in a cold second loop", launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 2, 0, 1>, t);
    }
    // ---- round 5, ONE counter-guided experiment: the next chunk's loads spread over the trip (LSP, cycle_kernel_lab.h)
    for (uint32_t g : {200u, 256u}) {
        char b_[160];
        snprintf(b_, sizeof b_, "queue   LSP 1: loads as two half-bursts (2 at the start, 2 behind word 1)   grid= %u", autogrid(65536, g));
        vs.push_back({0, b_, launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 1>, 65536, autogrid(65536, g), {}});
        snprintf(b_, sizeof b_, "queue   LSP 2: 2 words of chunk k+2 behind trip k's store burst            grid= %u", autogrid(65536, g));
        vs.push_back({0, b_, launch_queue<4, 1024, 0, 1, MODE_FULL, 18, 2, 1, 1, 2, 0, 1, 0, 2>, 65536, autogrid(65536, g), {}});
        snprintf(b_, sizeof b_, "queue   LSP 0: the product's single burst (lab form)                        grid= %u", autogrid(65536, g));
        vs.push_back({0, b_, launch_queue<4, 1024, 0, 1, MODE_FULL, 18>, 65536, autogrid(65536, g), {}});
    }
    vs.push_back({0, "queue   LSP 1 COPY-ONLY                                                          grid= 200", launch_queue<4, 1024, 0, 1, MODE_COPY, 18, 2, 1, 1, 2, 0, 1, 0, 1>, 65536, autogrid(65536, 200), {}});
    vs.push_back({0, "queue   LSP 2 COPY-ONLY                                                          grid= 200", launch_queue<4, 1024, 0, 1, MODE_COPY, 18, 2, 1, 1, 2, 0, 1, 0, 2>, 65536, autogrid(65536, 200), {}});
    vs.push_back({0, "queue   LSP 0 COPY-ONLY                                                          grid= 200", launch_queue<4, 1024, 0, 1, MODE_COPY, 18>, 65536, autogrid(65536, 200), {}});
    // TUNE_ONLY="token|token": keep the variants whose name contains one of the tokens (an experiment's rows interleaved with each
    // other only, in a fraction of the time the whole table takes)
    if (const char *only = getenv("TUNE_ONLY")) {
        std::vector<std::string> toks;
        std::string cur;
        for (const char *c = only;; ++c) {
            if (*c == '|' || *c == 0) {
                if (!cur.empty()) toks.push_back(cur);
                cur.clear();
                if (*c == 0) break;
            } else cur.push_back(*c);
        }
        std::vector<Variant> keep;
        for (auto &v : vs)
            for (auto &t : toks)
                if (v.name.find(t) != std::string::npos) {
                    keep.push_back(v);
                    break;
                }
        vs.swap(keep);
    }
    LabArgs a{};
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
            a.tail_chunks = v.tail;
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
    // validity pass.  (1) involution: two launches of each variant on the freshly filled buffer must give the fill pattern
    // back -- sees chunks that were skipped or done twice.  (2) keystream: ONE launch over zero bytes must leave the
    // reference keystream, byte for byte over the whole buffer -- sees wrong arithmetic, which (1) cannot (see above).
    unsigned long long *d_bad;
    CHECK(hipMalloc(&d_bad, 8));
    pin_reference_kernel(buf + n, n, st); // (the allocation has 1 MiB of slack behind the buffer)
    for (auto &v : vs) {
        if (v.name.find("compute") == 0) continue; // the compute-only ablation never stores
        a.tail_chunks = v.tail;
        if (v.name.find("copy") != 0 && v.name.find("COPY") == std::string::npos) { // (a copy-only ablation has no keystream)
            CHECK(hipMemsetAsync(buf, 0, n, st));
            a.body = buf + v.base_off;
            a.lead = (uint32_t)((uintptr_t)a.body & (v.chunk - 1));
            a.base_body = lcg::mulmod(base0, lcg::powmod(lcg::A, lcg::PERIOD - a.lead));
            a.stride_mul2 = 2u * lcg::powmod(lcg::A, ((uint64_t)v.grid * v.chunk) % lcg::PERIOD);
            v.launch(a, v.grid, st);
            v.ks_bad = keystream_mismatches(buf, n / 16 * 16, d_bad, st);
        }
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
               2.0 * n / med / 1e6, 2.0 * n / mn / 1e6,
               v.bad ? "   ** INVALID: chunks skipped or repeated, timing meaningless **" : v.ks_bad ? "   ** INVALID: WRONG KEYSTREAM **" : "");
    }
    return 0;
}
