// standin_launch.cpp -- the launch side of the CPU stand-in (tests/cpu_runtime_standin/hip/hip_runtime.h): cycle_kernel.h's interface
// implemented on the host.  A "launch" is queued on the stream's thread and does, from the launch PLAN alone
// (CycleArgs: head / body / tail pointers, the three base states, `lead`), what the kernel would do to the same
// bytes -- with the product's own Park-Miller arithmetic (lcg.h), byte by byte.  So the sanitizer runs check the
// host's planning (splits, jump-ahead states, alignment lead) as well as its memory and thread discipline.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>

#include "../../modulate_amd/csrc/cycle_feed_kernel.h"
#include "../../modulate_amd/csrc/cycle_kernel.h"
#include "../../modulate_amd/csrc/lcg.h"

namespace {
std::atomic<unsigned long long> g_collisions{0}, g_launches[kCycleVariants] = {};

struct Launch { CycleArgs a; int variant; };

// MODGPU_SHIM_SLOW=N (tests): the stand-in "GPU" takes N times as long over every span -- what a loaded box does to it anyway, on
// purpose: a test that only passes while the stand-in is quick depends on the clock, not on the library.
const int g_slow = [] {
    const char *v = std::getenv("MODGPU_SHIM_SLOW");
    const int x = v ? std::atoi(v) : 1;
    return x < 1 ? 1 : x;
}();
void span(uint8_t *p, uint64_t n, uint32_t state) // state = canonical state of p[0]
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t i = 0; i < n; ++i) {
        p[i] ^= (uint8_t)~state;
        state = lcg::mulmod(state, lcg::A);
    }
    if (g_slow > 1 && n >= 4096) std::this_thread::sleep_for((std::chrono::steady_clock::now() - t0) * (g_slow - 1));
}

void run(void *arg)
{
    Launch *l = static_cast<Launch *>(arg);
    const CycleArgs &a = l->a;
    span(a.head_ptr, a.head_n, a.base_head);
    // base_body is the state `lead` bytes before the body
    span(static_cast<uint8_t *>(a.body), a.body_words * lcg::WORD, lcg::mulmod(a.base_body, lcg::powmod(lcg::A, a.lead)));
    span(a.tail_ptr, a.tail_n, a.base_tail);
    g_launches[l->variant].fetch_add(1);
    delete l;
}
} // namespace

uint32_t modgpu_variant_chunk_bytes(int variant) { return variant == CYCLE_QUEUE ? 65536u : variant == CYCLE_LARGE ? 131072u : 4096u; }
uint32_t modgpu_variant_block(int variant) { return variant == CYCLE_SMALL ? 256u : 1024u; }
const char *modgpu_variant_kernel_name(int variant) { return variant == CYCLE_QUEUE ? "shim queue" : variant == CYCLE_LARGE ? "shim large" : "shim small"; }

hipError_t modgpu_launch_cycle(const CycleArgs &a, int variant, uint32_t, hipStream_t stream)
{
    shim::enqueue(stream, run, new Launch{a, variant});
    return hipSuccess;
}

// the work-queue shape: the same, part by part, from the table (start[] must tile the chunk index space).  The ticket pair is
// emulated: taken at the start of the launch, cleaned and signed off at its end; a launch that finds its pair taken counts a collision.
namespace {
std::atomic<unsigned long long> g_batch_launches{0}, g_batch_plan_errors{0};
void run_batch(void *arg)
{
    CycleQueueArgs *b = static_cast<CycleQueueArgs *>(arg);
    uint32_t expect = 0;
    if (!std::atomic_ref<uint32_t>(b->queue[0]).compare_exchange_strong(expect, 1u)) g_collisions.fetch_add(1); // another launch holds this pair
    std::this_thread::sleep_for(std::chrono::microseconds(200)); // a launch lasts a while: overlaps become likely
    const uint64_t chunk = modgpu_queue_chunk_bytes();
    uint64_t total = 0;
    for (uint32_t p = 0; p < b->n_parts; ++p) {
        const CycleQueuePart &P = b->part[p];
        const uint64_t body_bytes = P.end - P.lead, n_chunks = (P.end + chunk - 1) / chunk, first = P.lead != 0 ? 1 : 0;
        if (b->start[p] != total || (reinterpret_cast<uintptr_t>(P.body) & (chunk - 1)) != P.lead) g_batch_plan_errors.fetch_add(1);
        total += n_chunks > first ? n_chunks - first : 0;
        span(P.body - P.head_n, P.head_n, P.base_head);
        span(P.body, body_bytes, lcg::mulmod(P.base_body, lcg::powmod(lcg::A, P.lead)));
        span(P.body + body_bytes, P.tail_n, P.base_tail);
    }
    for (uint32_t p = b->n_parts; p <= (uint32_t)kCycleBatchMax; ++p)
        if (b->start[p] != total) g_batch_plan_errors.fetch_add(1);
    std::atomic_ref<uint32_t>(b->queue[0]).store(0u);
    if (b->queue_done) std::atomic_ref<uint32_t>(*b->queue_done).store(b->queue_seq, std::memory_order_release);
    if (b->n_parts > 1) g_batch_launches.fetch_add(1);
    g_launches[CYCLE_QUEUE].fetch_add(1);
    delete b;
}
} // namespace
uint32_t modgpu_queue_chunk_bytes() { return 65536u; }
uint32_t modgpu_queue_block() { return 1024u; }
const char *modgpu_queue_kernel_name() { return "shim queue"; }
hipError_t modgpu_launch_cycle_queue(const CycleQueueArgs &a, uint32_t, hipStream_t stream)
{
    shim::enqueue(stream, run_batch, new CycleQueueArgs(a));
    return hipSuccess;
}
// the host-fed kernel: the same protocol on the stream's thread -- chunk after chunk, wait for `ready` (or `abort`, or patience), cycle the
// chunk in its slot from the launch arguments alone, mark it `done`.  Runs WHILE the library's pipelines copy in and out, as the kernel does.
namespace {
std::atomic<unsigned long long> g_feed_launches{0}, g_feed_gave_up{0};
// modgpu_shim_wedge_next_feed(1): the next host-fed "kernel" stops responding half-way -- it neither finishes its chunks nor ends, whatever
// the host's abort word says (a workgroup stuck at a barrier) -- until modgpu_shim_release_wedged() lets the stream's thread go.
std::atomic<int> g_wedge_next{0};
std::atomic<bool> g_wedge_release{false};
void run_feed(void *arg)
{
    CycleFeedArgs *a = static_cast<CycleFeedArgs *>(arg);
    const uint64_t chunks = (a->n + a->chunk_bytes - 1) / a->chunk_bytes;
    bool gave_up = false;
    const bool wedge = g_wedge_next.exchange(0) != 0;
    for (uint64_t c = 0; c < chunks && !gave_up; ++c) {
        if (wedge && c == chunks / 2) {
            while (!g_wedge_release.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            delete a; // (nothing of the call is touched any more: the library has long abandoned it)
            return;
        }
        const auto since = std::chrono::steady_clock::now();
        while (std::atomic_ref<const uint32_t>(a->ready[c]).load(std::memory_order_acquire) == 0u) {
            const double waited_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - since).count();
            if (std::atomic_ref<const uint32_t>(*a->abort).load(std::memory_order_acquire) != 0u || waited_s * 1e8 > (double)a->patience_ticks) {
                gave_up = true;
                break;
            }
            std::this_thread::yield();
        }
        if (gave_up) break;
        const uint64_t pos = c * a->chunk_bytes, len = std::min<uint64_t>(a->chunk_bytes, a->n - pos);
        uint8_t *slot = a->slot[(c % a->pipes) * 2 + (c / a->pipes) % 2];
        span(slot, len, lcg::mulmod(a->base, lcg::powmod(lcg::A, pos % lcg::PERIOD)));
        std::atomic_ref<uint32_t>(a->done[c]).store(1u, std::memory_order_release);
    }
    if (gave_up) {
        g_feed_gave_up.fetch_add(1);
        std::atomic_ref<uint32_t>(a->work[1]).fetch_add(1u);
    }
    g_feed_launches.fetch_add(1);
    delete a;
}
} // namespace
uint32_t modgpu_feed_block() { return 256u; }
const char *modgpu_feed_kernel_name() { return "shim feed"; }
hipError_t modgpu_launch_cycle_feed(const CycleFeedArgs &a, uint32_t, hipStream_t stream)
{
    shim::enqueue(stream, run_feed, new CycleFeedArgs(a));
    return hipSuccess;
}
extern "C" void modgpu_shim_wedge_next_feed(int on) { g_wedge_next.store(on ? 1 : 0); }
extern "C" void modgpu_shim_release_wedged(void) { g_wedge_release.store(true, std::memory_order_release); }
extern "C" unsigned long long modgpu_shim_feed_launches(void) { return g_feed_launches.load(); }
extern "C" unsigned long long modgpu_shim_feed_gave_up(void) { return g_feed_gave_up.load(); }

extern "C" unsigned long long modgpu_shim_batch_launches(void) { return g_batch_launches.load(); }
extern "C" unsigned long long modgpu_shim_batch_plan_errors(void) { return g_batch_plan_errors.load(); }

extern "C" unsigned long long modgpu_shim_pair_collisions(void) { return g_collisions.load(); }
extern "C" unsigned long long modgpu_shim_launches(int variant) { return variant >= 0 && variant < kCycleVariants ? g_launches[variant].load() : 0; }
