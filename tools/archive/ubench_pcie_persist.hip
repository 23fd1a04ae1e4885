// tools/ubench_pcie_persist.hip [MiB=64] [reps=12] -- a LAB measurement, not the product: what would the pageable (staged) host route
// reach if ONE kernel per call took the chunks as the host marks them copied in, instead of one kernel launch per chunk?
// (the chunk kernels' ramp-up and drain is what separates the staged route from its schedule bound.)
//
//   launches   the product's schedule in miniature: P host threads, each copy-in -> kernel across PCIe on the chunk (grid 32 x 256, one
//              16-byte word per lane per trip, 4 shared lanes) -> wait -> copy-out, double-buffered per thread
//   persist    the same threads and copies; ONE kernel of G workgroups launched when the call starts.  A workgroup draws a ticket (a 32 KiB
//              piece), waits until the host has marked the piece's chunk `ready` (a word in page-locked host memory), does the piece,
//              counts it; the workgroup that completes a chunk marks it `done` in host memory, which the host thread polls.
// The kernel only XORs a constant (the link is the bound, not the keystream).  Every wait in the kernel has a deadline by the wall clock
// (s_memrealtime): a host that never marks a chunk ends the kernel after 50 ms, it cannot hang.  Results are checked whole.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr uint32_t kXor = 0x5A5A5A5Au;
typedef uint32_t word4 __attribute__((ext_vector_type(4))); // (the nontemporal builtins want a native vector)
// the product's cache policy for the shape it runs across PCIe (cycle_kernel_impl.h): loads non-temporal, stores write-through (sc1)
__device__ inline void store_sc1(word4 *p, word4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
constexpr uint32_t kPieceBytes = 32u << 10, kPieceWords = kPieceBytes / 16;

struct PersistArgs {
    word4 *slot;           // the staging buffer as the device addresses it
    const uint32_t *ready; // [chunks] host memory, written by the host
    uint32_t *done;        // [chunks] host memory, written by the kernel
    uint32_t *cnt;         // [chunks] device memory, zero at launch: pieces of the chunk finished
    uint32_t *ticket;      // device memory, zero at launch
    uint32_t *gave_up;     // device memory: workgroups that met the deadline
    uint32_t n_tickets, tickets_per_chunk;
    uint32_t pipes, whole; // chunk k lives in slot (k % pipes) * 2 + (k / pipes) % 2 of tickets_per_chunk pieces (whole != 0: at its own place in an n-byte staging buffer)
    uint64_t deadline_ticks; // of the 100 MHz wall clock
};

// (Shape of the loop: everything only thread 0 does sits in ONE region at the top of a trip, in front of the first barrier -- counting
//  the piece of the trip before, then drawing and waiting for the next.  A second thread-0 region behind the last barrier of the trip had
//  the compiler send lanes 1..63 of wave 0 round the back edge on their own: that wave then passed the barrier twice per trip, the other
//  waves once, and the workgroup hung.)
template <int U>
__global__ __launch_bounds__(256) void persist_kernel(PersistArgs a)
{
    __shared__ uint32_t s_t, s_ok;
    const uint64_t t0 = wall_clock64();
    uint32_t counted = 0xFFFFFFFFu; // thread 0: the chunk of the piece this workgroup has finished and not yet counted
    for (;;) {
        if (threadIdx.x == 0) {
            if (counted != 0xFFFFFFFFu &&
                __hip_atomic_fetch_add(&a.cnt[counted], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == a.tickets_per_chunk - 1)
                __hip_atomic_store(&a.done[counted], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t t = atomicAdd(a.ticket, 1u);
            uint32_t ok = 1;
            if (t < a.n_tickets) {
                const uint32_t c = t / a.tickets_per_chunk;
                counted = c;
                while (__hip_atomic_load(&a.ready[c], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
                    if (wall_clock64() - t0 > a.deadline_ticks) { // the exit every waiting wave reaches
                        ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            s_t = t;
            s_ok = ok;
        }
        __syncthreads();
        const uint32_t t = s_t, ok = s_ok;
        if (t >= a.n_tickets || !ok) {
            if (!ok && threadIdx.x == 0) atomicAdd(a.gave_up, 1u);
            break;
        }
        const uint32_t ck = t / a.tickets_per_chunk, piece = t - ck * a.tickets_per_chunk;
        const uint32_t slot = a.whole ? ck : (ck % a.pipes) * 2u + (ck / a.pipes) % 2u;
        word4 *p = a.slot + ((uint64_t)slot * a.tickets_per_chunk + piece) * kPieceWords;
        for (uint32_t i = threadIdx.x; i < kPieceWords; i += 256 * U) {
            word4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&p[i + u * 256]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] ^= kXor;
                store_sc1(&p[i + u * 256], v[u]);
            }
        }
        __threadfence_system(); // this wave's stores have reached host memory ...
        __syncthreads();        // ... and every wave's have, before thread 0 counts the piece at the top of the next trip
    }
}

__global__ __launch_bounds__(256) void chunk_kernel(word4 *p, uint32_t words)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < words; i += gridDim.x * 256) {
        word4 v = __builtin_nontemporal_load(&p[i]);
        v ^= kXor;
        store_sc1(&p[i], v);
    }
}

__attribute__((target("avx2"))) static void copy_nt(uint8_t *dst, const uint8_t *src, size_t n) // n a multiple of 128, dst 32-byte aligned
{
    for (size_t i = 0; i < n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32)),
                      c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c);
        _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    _mm_sfence();
}

constexpr int P = 8, kLanes = 4;
struct Ctx {
    size_t n = 0, chunk = 0, chunks = 0;
    uint8_t *src = nullptr, *dst = nullptr, *slot = nullptr;
    word4 *slot_dev = nullptr;
    uint32_t *ready = nullptr, *done = nullptr, *ready_dev = nullptr, *done_dev = nullptr, *dev_words = nullptr; // dev_words: cnt[chunks], ticket, gave_up
    hipStream_t main_st = nullptr, lane[kLanes] = {};
    std::vector<hipEvent_t> ev;
    int mode = 0, grid = 32; // mode 0 launches, 1 persist
    bool whole = false;      // staging as large as the buffer (every chunk its own place) instead of two slots per host thread
    size_t slot_of(size_t k) const { return whole ? k : (k % 8) * 2 + (k / 8) % 2; }
    std::atomic<int> failed{0};
};

static bool g_dbg = false;
static double g_t0 = 0;
#define DBG(...) do { if (g_dbg) { fprintf(stderr, "[%.0f us] ", now_us() - g_t0); fprintf(stderr, __VA_ARGS__); } } while (0)
static void worker(Ctx &c, int w)
{
    std::vector<size_t> mine;
    for (size_t k = (size_t)w; k < c.chunks; k += P) mine.push_back(k);
    auto in = [&](size_t k) {
        copy_nt(c.slot + c.slot_of(k) * c.chunk, c.src + k * c.chunk, c.chunk);
        if (c.mode == 1) {
            __atomic_store_n(&c.ready[k], 1u, __ATOMIC_RELEASE);
        } else {
            hipStream_t st = c.lane[k % kLanes];
            hipLaunchKernelGGL(chunk_kernel, dim3(c.grid), dim3(256), 0, st, c.slot_dev + c.slot_of(k) * c.chunk / 16, (uint32_t)(c.chunk / 16));
            CHECK(hipEventRecord(c.ev[k], st));
        }
    };
    auto wait = [&](size_t k) {
        if (c.mode == 1) {
            const double t0 = now_us();
            while (__atomic_load_n(&c.done[k], __ATOMIC_ACQUIRE) == 0u) {
                _mm_pause();
                if (now_us() - t0 > 200000.0) { // the kernel has given up, or never ran
                    c.failed.store(1);
                    return;
                }
            }
        } else {
            CHECK(hipEventSynchronize(c.ev[k]));
        }
    };
    for (size_t j = 0; j < mine.size() && j < 2; ++j) in(mine[j]);
    if (w == 0) DBG("worker 0: first two chunks in\n");
    for (size_t j = 0; j < mine.size(); ++j) {
        wait(mine[j]);
        if (w == 0) DBG("worker 0: chunk %zu waited for (failed %d)\n", mine[j], c.failed.load());
        if (c.failed.load()) return;
        copy_nt(c.dst + mine[j] * c.chunk, c.slot + c.slot_of(mine[j]) * c.chunk, c.chunk);
        if (j + 2 < mine.size()) in(mine[j + 2]);
    }
}

// parked threads with a spinning start line (the product parks its workers on a condition variable; the wake-up is not what is measured here)
struct Crew {
    std::atomic<int> gen{0}, finished{0};
    std::atomic<bool> quit{false};
    Ctx *ctx = nullptr;
    std::vector<std::thread> th;
    void start()
    {
        for (int w = 1; w < P; ++w)
            th.emplace_back([this, w] {
                int seen = 0;
                for (;;) {
                    while (gen.load(std::memory_order_acquire) == seen && !quit.load()) _mm_pause();
                    if (quit.load()) return;
                    ++seen;
                    worker(*ctx, w);
                    finished.fetch_add(1, std::memory_order_release);
                }
            });
    }
    void run(Ctx &c)
    {
        ctx = &c;
        finished.store(0);
        gen.fetch_add(1, std::memory_order_release);
        worker(c, 0);
        while (finished.load(std::memory_order_acquire) < P - 1) _mm_pause();
    }
    void stop()
    {
        quit.store(true);
        for (auto &t : th) t.join();
    }
};

template <int U>
static void launch_persist(Ctx &c)
{
    PersistArgs a{};
    a.slot = c.slot_dev;
    a.ready = c.ready_dev;
    a.done = c.done_dev;
    a.cnt = c.dev_words;
    a.ticket = c.dev_words + c.chunks;
    a.gave_up = c.dev_words + c.chunks + 1;
    a.tickets_per_chunk = (uint32_t)(c.chunk / kPieceBytes);
    a.n_tickets = (uint32_t)(c.n / kPieceBytes);
    a.deadline_ticks = 5000000ull; // 50 ms
    a.pipes = (uint32_t)P;
    a.whole = c.whole ? 1u : 0u;
    hipLaunchKernelGGL(persist_kernel<U>, dim3(c.grid), dim3(256), 0, c.main_st, a);
}

static double one_call(Ctx &c, Crew &crew, int unroll)
{
    const double t0 = now_us();
    g_t0 = t0;
    if (c.mode == 1) {
        std::memset(c.ready, 0, c.chunks * 4);
        std::memset(c.done, 0, c.chunks * 4);
        CHECK(hipMemsetAsync(c.dev_words, 0, (c.chunks + 2) * 4, c.main_st));
        if (unroll == 1) launch_persist<1>(c);
        else if (unroll == 2) launch_persist<2>(c);
        else launch_persist<4>(c);
    }
    DBG("launched\n");
    crew.run(c);
    DBG("crew done\n");
    if (c.mode == 1) CHECK(hipStreamSynchronize(c.main_st));
    DBG("kernel done\n");
    return now_us() - t0;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 64;
    const int reps = argc > 2 ? atoi(argv[2]) : 12;
    g_dbg = getenv("PERSIST_DEBUG") != nullptr;
    Ctx c;
    c.n = mib << 20;
    c.src = (uint8_t *)aligned_alloc(4096, c.n);
    c.dst = (uint8_t *)aligned_alloc(4096, c.n);
    for (size_t i = 0; i < c.n; i += 8) *(uint64_t *)(c.src + i) = i * 0x9E3779B97F4A7C15ull;
    std::memset(c.dst, 0, c.n);
    CHECK(hipHostMalloc((void **)&c.slot, c.n, hipHostMallocPortable | hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void **)&c.slot_dev, c.slot, 0));
    const size_t max_chunks = c.n / (128u << 10);
    CHECK(hipHostMalloc((void **)&c.ready, max_chunks * 4, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostMalloc((void **)&c.done, max_chunks * 4, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer((void **)&c.ready_dev, c.ready, 0));
    CHECK(hipHostGetDevicePointer((void **)&c.done_dev, c.done, 0));
    CHECK(hipMalloc((void **)&c.dev_words, (max_chunks + 2) * 4));
    CHECK(hipStreamCreateWithFlags(&c.main_st, hipStreamNonBlocking));
    for (auto &l : c.lane) CHECK(hipStreamCreateWithFlags(&l, hipStreamNonBlocking));
    c.ev.resize(max_chunks);
    for (auto &e : c.ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (const char *t = getenv("PERSIST_TEST")) { // the kernel alone: A = every chunk ready beforehand, B = none ever (the deadline must end it)
        c.chunk = 1u << 20;
        c.chunks = c.n / c.chunk;
        c.whole = true;
        std::memset(c.ready, *t == 'A' ? 1 : 0, c.chunks * 4);
        std::memset(c.done, 0, c.chunks * 4);
        std::memcpy(c.slot, c.src, c.n);
        const double t0 = now_us();
        CHECK(hipMemsetAsync(c.dev_words, 0, (c.chunks + 2) * 4, c.main_st));
        launch_persist<1>(c);
        CHECK(hipGetLastError());
        fprintf(stderr, "test %c: launched after %.0f us\n", *t, now_us() - t0);
        CHECK(hipStreamSynchronize(c.main_st));
        uint32_t words[3] = {};
        CHECK(hipMemcpy(words, c.dev_words + c.chunks - 1, 12, hipMemcpyDeviceToHost));
        fprintf(stderr, "test %c: kernel ended after %.0f us; cnt[last] %u ticket %u gave_up %u done[0] %u done[last] %u\n", *t, now_us() - t0, words[0], words[1], words[2], c.done[0], c.done[c.chunks - 1]);
        return 0;
    }
    Crew crew;
    crew.start();
    printf("== %zu MiB pageable -> page-locked staging -> kernel across PCIe -> pageable, %d host threads, %d calls per row (2 untimed first); GB/s of payload\n", mib, P, reps);
    printf("   %-10s %9s %6s %7s   %8s %8s   %s\n", "mode", "chunk KiB", "grid", "unroll", "best", "median", "check");
    struct Row { int mode; size_t chunk_kib; int grid, unroll; bool whole = false; };
    std::vector<Row> rows = {{0, 1024, 32, 1}, {0, 512, 32, 1}, {0, 2048, 32, 1}};
    if (getenv("PERSIST_DEBUG")) rows = {{1, 1024, 32, 1}};
    for (size_t ck : {256, 512, 1024})
        for (int g : {16, 32, 64})
            for (int u : {1, 4}) rows.push_back({1, ck, g, u});
    rows.push_back({0, 1024, 32, 1, true}); // staging as large as the buffer, for comparison
    rows.push_back({1, 512, 64, 1, true});
    rows.push_back({0, 1024, 32, 1}); // the first row again: drift
    for (const Row &r : rows) {
        c.mode = r.mode;
        c.chunk = r.chunk_kib << 10;
        c.grid = r.grid;
        c.whole = r.whole;
        if (c.n % c.chunk || c.n / c.chunk < 2 * (size_t)P) continue;
        c.chunks = c.n / c.chunk;
        std::vector<double> us;
        for (int k = 0; k < reps + 2 && !c.failed.load(); ++k) {
            const double t = one_call(c, crew, r.unroll);
            if (getenv("PERSIST_DEBUG")) fprintf(stderr, "  %s chunk %zu KiB grid %d unroll %d call %d: %.1f us\n", r.mode ? "persist" : "launches", r.chunk_kib, r.grid, r.unroll, k, t);
            if (k >= 2) us.push_back(t);
        }
        uint32_t gave_up = 0;
        if (c.mode == 1) CHECK(hipMemcpy(&gave_up, c.dev_words + c.chunks + 1, 4, hipMemcpyDeviceToHost));
        if (c.failed.load() || gave_up) {
            printf("   %-10s %9zu %6d %7d   the kernel gave up (%u workgroups at the deadline) or a host wait timed out: stopping\n", r.mode ? "persist" : "launches", r.chunk_kib, r.grid, r.unroll, gave_up);
            break;
        }
        size_t bad = 0;
        for (size_t i = 0; i < c.n; i += 8) bad += *(uint64_t *)(c.dst + i) != (*(uint64_t *)(c.src + i) ^ 0x5A5A5A5A5A5A5A5Aull);
        std::sort(us.begin(), us.end());
        printf("   %-10s %9zu %6d %7d   %8.2f %8.2f   %s%s\n", r.mode ? "persist" : "launches", r.chunk_kib, r.grid, r.unroll, c.n / us.front() / 1e3, c.n / us[us.size() / 2] / 1e3,
               bad ? "MISMATCH" : "ok", r.whole ? "   (staging as large as the buffer)" : "");
        fflush(stdout);
        std::memset(c.dst, 0, c.n);
    }
    crew.stop();
    CHECK(hipDeviceSynchronize());
    return 0;
}
