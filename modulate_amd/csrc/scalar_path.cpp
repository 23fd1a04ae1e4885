// scalar_path.cpp -- the library's own HOST implementation of the cipher: for machines without a
// usable GPU (the reference's Cycle "cannot fail", Modulate/CEncryptionCycler.cpp:4-14) and for the
// header-sized buffers the reference's three call sites pass (CArk.cpp:338-339, 1135-1136,
// Modulate.cpp:485-486), where a kernel launch + wait costs more than the arithmetic.
//
// Product code: built from lcg.h's closed form  s_i = a^(i+1) * key mod m  like the kernel, not
// from the reference's Schrage step, and it shares nothing with the checker under oracle/.  No HIP
// in this file, so the sanitizer builds compile it as is.
//
// Layout mirrors the GPU's lane-words: W independent byte states S_j = S * a^j, each advanced by a^W
// per W-byte block, so there is no serial dependency between neighbouring bytes.  Three bodies, picked
// once per process from what the CPU offers (MODGPU_HOST_ISA=generic|avx2|avx512 overrides):
//   generic  W = 16, plain C++ (the compiler keeps the sixteen 32x32->64 multiplies in registers)
//   avx2     W = 32: eight ymm registers of four states (vpmuludq), Mersenne fold, packed to 32 bytes
//   avx512   W = 64: the same with zmm registers (AVX-512 F + BW)
// Buffers of a few MiB and up are cut into contiguous spans, one per host thread (every span jumps
// to its own position); the threads are parked workers, started on first use (see HostPool below).
#include "scalar_path.h"

#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "lcg.h"

namespace {

// x, y < 2^31  ->  x*y mod m, in [0, m).  2^31 == 1 (mod m): fold the 62-bit product once.
inline uint32_t mulmod_fold(uint32_t x, uint32_t y)
{
    uint64_t p = (uint64_t)x * y;
    uint32_t r = (uint32_t)(p & lcg::M) + (uint32_t)(p >> 31); // < 2m
    return r >= lcg::M ? r - lcg::M : r;
}

void span_generic(uint8_t *buf, uint64_t n, uint32_t key_res, uint64_t pos)
{
    constexpr int W = lcg::WORD;
    const uint32_t step = lcg::mulmod(lcg::kBytePow.v[W - 1], lcg::A); // a^16
    uint32_t s[W];
    s[0] = lcg::state_residue(key_res, pos);
    for (int j = 1; j < W; ++j) s[j] = mulmod_fold(s[j - 1], lcg::A);
    uint64_t i = 0;
    for (; i + W <= n; i += W) {
        uint8_t ks[W];
        for (int j = 0; j < W; ++j) {
            ks[j] = (uint8_t)~s[j]; // states are canonical and never 0 here (key_res != 0, m prime)
            s[j] = mulmod_fold(s[j], step);
        }
        for (int j = 0; j < W; ++j) buf[i + j] ^= ks[j];
    }
    for (int j = 0; i < n; ++i, ++j) buf[i] ^= (uint8_t)~s[j];
}

// Vector bodies.  R registers of L 64-bit lanes hold R*L = W states in the lanes' low halves (vpmuludq
// multiplies exactly those).  Register pairs are merged into 32-bit lanes, masked to the low byte and
// narrowed twice with the in-lane pack instructions; the packs interleave, so state (register 2k+o,
// lane l) ends up at byte  16*(l/2) + 4*k + 2*(l&1) + o  of the block -- the states are simply
// initialised in that order.
inline int packed_byte(int reg, int lane) { return 16 * (lane / 2) + 4 * (reg / 2) + 2 * (lane & 1) + (reg & 1); }

__attribute__((target("avx2"))) void span_avx2(uint8_t *buf, uint64_t n, uint32_t key_res, uint64_t pos)
{
    constexpr int W = 32, R = 8, L = 4;
    uint32_t s[W];
    s[0] = lcg::state_residue(key_res, pos);
    for (int j = 1; j < W; ++j) s[j] = mulmod_fold(s[j - 1], lcg::A);
    const uint32_t step = lcg::powmod(lcg::A, W);
    alignas(32) uint64_t init[R][L];
    for (int r = 0; r < R; ++r)
        for (int l = 0; l < L; ++l) init[r][l] = s[packed_byte(r, l)];
    __m256i v[R];
    for (int r = 0; r < R; ++r) v[r] = _mm256_load_si256(reinterpret_cast<const __m256i *>(init[r]));
    const __m256i vstep = _mm256_set1_epi64x(step), vm = _mm256_set1_epi64x(lcg::M), vff = _mm256_set1_epi32(0xFF),
                  ones = _mm256_set1_epi8((char)0xFF);
    uint64_t i = 0;
    for (; i + W <= n; i += W) {
        __m256i c[R / 2];
        for (int k = 0; k < R / 2; ++k)
            c[k] = _mm256_and_si256(_mm256_or_si256(v[2 * k], _mm256_slli_epi64(v[2 * k + 1], 32)), vff);
        const __m256i state_bytes = _mm256_packus_epi16(_mm256_packus_epi32(c[0], c[1]), _mm256_packus_epi32(c[2], c[3]));
        __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(buf + i));
        d = _mm256_xor_si256(_mm256_xor_si256(d, state_bytes), ones); // keystream byte = ~low8(state)
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(buf + i), d);
        for (int r = 0; r < R; ++r) {
            const __m256i p = _mm256_mul_epu32(v[r], vstep);
            const __m256i x = _mm256_add_epi64(_mm256_and_si256(p, vm), _mm256_srli_epi64(p, 31)); // < 2m, == p (mod m)
            v[r] = _mm256_min_epu32(x, _mm256_sub_epi32(x, vm)); // canonical (x != m: m is prime, no zero factors)
        }
    }
    if (i < n) { // ragged end: the next block's bytes, as many as are left
        alignas(32) uint64_t last[R][L];
        for (int r = 0; r < R; ++r) _mm256_store_si256(reinterpret_cast<__m256i *>(last[r]), v[r]);
        uint8_t ks[W];
        for (int r = 0; r < R; ++r)
            for (int l = 0; l < L; ++l) ks[packed_byte(r, l)] = (uint8_t)~last[r][l];
        for (int j = 0; i < n; ++i, ++j) buf[i] ^= ks[j];
    }
}

__attribute__((target("avx512f,avx512bw"))) void span_avx512(uint8_t *buf, uint64_t n, uint32_t key_res, uint64_t pos)
{
    constexpr int W = 64, R = 8, L = 8;
    uint32_t s[W];
    s[0] = lcg::state_residue(key_res, pos);
    for (int j = 1; j < W; ++j) s[j] = mulmod_fold(s[j - 1], lcg::A);
    const uint32_t step = lcg::powmod(lcg::A, W);
    alignas(64) uint64_t init[R][L];
    for (int r = 0; r < R; ++r)
        for (int l = 0; l < L; ++l) init[r][l] = s[packed_byte(r, l)];
    __m512i v[R];
    for (int r = 0; r < R; ++r) v[r] = _mm512_load_si512(init[r]);
    const __m512i vstep = _mm512_set1_epi64(step), vm = _mm512_set1_epi64(lcg::M), vff = _mm512_set1_epi32(0xFF),
                  ones = _mm512_set1_epi8((char)0xFF);
    uint64_t i = 0;
    for (; i + W <= n; i += W) {
        __m512i c[R / 2];
        for (int k = 0; k < R / 2; ++k)
            c[k] = _mm512_and_si512(_mm512_or_si512(v[2 * k], _mm512_slli_epi64(v[2 * k + 1], 32)), vff);
        const __m512i state_bytes = _mm512_packus_epi16(_mm512_packus_epi32(c[0], c[1]), _mm512_packus_epi32(c[2], c[3]));
        __m512i d = _mm512_loadu_si512(buf + i);
        d = _mm512_ternarylogic_epi64(d, state_bytes, ones, 0x96); // d ^ state ^ 0xFF
        _mm512_storeu_si512(buf + i, d);
        for (int r = 0; r < R; ++r) {
            const __m512i p = _mm512_mul_epu32(v[r], vstep);
            const __m512i x = _mm512_add_epi64(_mm512_and_si512(p, vm), _mm512_srli_epi64(p, 31));
            v[r] = _mm512_min_epu32(x, _mm512_sub_epi32(x, vm));
        }
    }
    if (i < n) {
        alignas(64) uint64_t last[R][L];
        for (int r = 0; r < R; ++r) _mm512_store_si512(last[r], v[r]);
        uint8_t ks[W];
        for (int r = 0; r < R; ++r)
            for (int l = 0; l < L; ++l) ks[packed_byte(r, l)] = (uint8_t)~last[r][l];
        for (int j = 0; i < n; ++i, ++j) buf[i] ^= ks[j];
    }
}

using SpanFn = void (*)(uint8_t *, uint64_t, uint32_t, uint64_t);
constexpr SpanFn kSpan[MODGPU_ISA_COUNT] = {span_generic, span_avx2, span_avx512};
constexpr const char *kIsaName[MODGPU_ISA_COUNT] = {"generic", "avx2", "avx512"};

bool isa_usable(int isa)
{
    switch (isa) {
    case MODGPU_ISA_GENERIC: return true;
    case MODGPU_ISA_AVX2: return __builtin_cpu_supports("avx2");
    case MODGPU_ISA_AVX512: return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    default: return false;
    }
}

int pick_isa()
{
    static const int v = [] {
        if (const char *e = std::getenv("MODGPU_HOST_ISA"))
            for (int i = 0; i < MODGPU_ISA_COUNT; ++i)
                if (std::strcmp(e, kIsaName[i]) == 0 && isa_usable(i)) return i;
        for (int i = MODGPU_ISA_COUNT - 1; i > 0; --i)
            if (isa_usable(i)) return i;
        return (int)MODGPU_ISA_GENERIC;
    }();
    return v;
}

// MODGPU_HOST_THREADS (read once): most host threads one call may use; 0 / unset = min(hardware, 32)
unsigned max_threads()
{
    static const unsigned v = [] {
        const char *e = std::getenv("MODGPU_HOST_THREADS");
        int x = e ? std::atoi(e) : 0;
        unsigned hw = std::thread::hardware_concurrency();
        return x > 0 ? (unsigned)std::min(x, 256) : std::max(1u, std::min(hw, 32u));
    }();
    return v;
}

// CPUs' worth of run time the process's control group may use (cgroup v2 cpu.max, v1 cfs quota / period); 0 = no limit
// known.  A container given 16 of a host's 256 CPUs sees all 256 in its affinity mask: 32 threads there are throttled by
// the bandwidth controller in bursts, which is what made round 3's threaded rows erratic (34 GB/s between 56 and 142 at
// 64 MiB on the MI355X pool's 16-CPU share).  Read once: quotas do not change under a running process as a rule.
unsigned cgroup_cpu_limit()
{
    static const unsigned v = [] {
#ifdef MODGPU_TESTING_HOOKS // (testing flavour only: MODGPU_HOST_CGROUP=0 = do not look -- short bursts are not throttled; a measurement switch)
        if (const char *e = std::getenv("MODGPU_HOST_CGROUP"))
            if (std::strcmp(e, "0") == 0) return 0u;
#endif
        auto read2 = [](const char *path, long long *a, long long *b) {
            FILE *f = std::fopen(path, "r");
            if (!f) return 0;
            char w0[32] = {}, w1[32] = {};
            const int n = std::fscanf(f, "%31s %31s", w0, w1);
            std::fclose(f);
            if (n >= 1) *a = std::strcmp(w0, "max") == 0 ? -1 : std::atoll(w0);
            if (n >= 2) *b = std::atoll(w1);
            return n;
        };
        long long quota = -1, period = 0, dummy = 0;
        if (read2("/sys/fs/cgroup/cpu.max", &quota, &period) < 2) { // v1
            if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &quota, &dummy) < 1 || read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &period, &dummy) < 1) return 0u;
        }
        if (quota <= 0 || period <= 0) return 0u;
        return (unsigned)std::max<long long>(1, (quota + period - 1) / period);
    }();
    return v;
}

// The CPUs the CALLING thread may run on right now (not a snapshot from the first call -- a process may narrow
// its affinity later, and workers must not be placed outside the mask the caller was given).
std::vector<int> allowed_cpus()
{
    std::vector<int> cpus;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) != 0) return cpus;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &set)) cpus.push_back(c);
    return cpus;
}
// MODGPU_HOST_SPREAD=1 (testing flavour): bind each worker to a CPU of its own for the length of its span.  OFF by default since round 4: the
// workers are parked threads now, which the scheduler wakes on idle CPUs all over the caller's mask, and for this memory-bound
// loop that is the better placement -- 16 unbound workers reach 113-180 GB/s at 16 MiB ... 1 GiB on the MI355X node's EPYC 9575F,
// 16 workers bound to the CPUs next to the caller's 79-110 (neighbouring CPU numbers are cores of one or two CCDs, which share
// a link into the memory fabric; profiles/r04_small_call_crossover.txt).  Round 3 bound them because its threads were started
// per call, and short-lived threads the scheduler is left to place can sit on one CPU for their whole life (measured in a VM).
bool spread_enabled()
{
#ifdef MODGPU_TESTING_HOOKS // (a measurement switch of the testing flavour; the shipped library never binds its workers)
    static const bool v = [] {
        const char *e = std::getenv("MODGPU_HOST_SPREAD");
        return e && std::strcmp(e, "1") == 0;
    }();
    return v;
#else
    return false;
#endif
}

// ---- parked workers ---------------------------------------------------------------------------------------------------------
// One call's spans, drawn by the caller and by parked workers until none is left (round 3 started up to 31 std::threads per
// call: ~30 us each, a millisecond for a 64 MiB buffer whose arithmetic takes a fifth of that).  With MODGPU_HOST_SPREAD=1 worker k of a
// call binds itself, for the length of its span, to the k-th CPU after the caller's in the CALLER's current affinity mask, and
// takes whatever the next call hands it; by default workers stay where the scheduler wakes them (see spread_enabled).  The
// caller's own thread is never re-bound.
struct SpanCall {
    SpanFn span;
    uint8_t *buf;
    uint64_t n, per, count;
    uint32_t key_res;
    uint64_t pos;
    std::vector<int> cpus; // per span index: CPU for a WORKER that draws it (-1 / empty: anywhere in the caller's mask)
    cpu_set_t caller_mask; // what the caller may run on (a worker left bound to one CPU by an earlier call is widened to it)
    bool have_mask = false;
    std::atomic<uint64_t> next{0};
    std::mutex mu;
    std::condition_variable cv;
    uint64_t finished = 0;
    void help(bool worker)
    {
        for (;;) {
            const uint64_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= count) return;
            if (worker && i < cpus.size() && cpus[i] >= 0) {
                cpu_set_t one;
                CPU_ZERO(&one);
                CPU_SET(cpus[i], &one);
                (void)sched_setaffinity(0, sizeof one, &one); // this worker thread only; best effort
            } else if (worker && have_mask) {
                (void)sched_setaffinity(0, sizeof caller_mask, &caller_mask);
            }
            const uint64_t off = i * per;
            span(buf + off, std::min(per, n - off), key_res, pos + off % lcg::PERIOD);
            std::lock_guard<std::mutex> lock(mu);
            if (++finished == count) cv.notify_all();
        }
    }
};
struct HostPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<SpanCall>> requests;
    unsigned workers = 0, parked = 0;
};
HostPool *g_pool = new HostPool; // never destroyed: parked workers wait on it for the life of the process
std::atomic<unsigned long long> g_pool_threads{0};
// fork(): the child has the calling thread only -- the parked workers are gone, but the pool would still count them (and its
// mutex may have been held by one of them at the moment of the fork): the child starts with a fresh, empty pool (ADVICE r4;
// Python's multiprocessing forks by default).  The old pool is leaked on purpose.
const int g_pool_atfork = ::pthread_atfork(nullptr, nullptr, [] { g_pool = new HostPool; });

void pool_worker()
{
    std::unique_lock<std::mutex> lock(g_pool->mu);
    for (;;) {
        ++g_pool->parked;
        g_pool->cv.wait(lock, [] { return !g_pool->requests.empty(); });
        --g_pool->parked;
        std::shared_ptr<SpanCall> call = std::move(g_pool->requests.front());
        g_pool->requests.pop_front();
        lock.unlock();
        call->help(true);
        call.reset();
        lock.lock();
    }
}

} // namespace

int modgpu_scalar_isa() { return pick_isa(); }
const char *modgpu_scalar_isa_name(int isa) { return isa >= 0 && isa < MODGPU_ISA_COUNT ? kIsaName[isa] : "?"; }
bool modgpu_scalar_isa_usable(int isa) { return isa_usable(isa); }

unsigned modgpu_scalar_threads_for(uint64_t n)
{
    // a span is worth a thread from ~2 MiB (a few hundred microseconds of work against ~20 us to wake a parked worker)
    constexpr uint64_t kSpanMin = 2ull << 20;
    uint64_t t = std::min<uint64_t>(max_threads(), n / kSpanMin);
    if (const unsigned lim = cgroup_cpu_limit()) t = std::min<uint64_t>(t, lim);
    if (t > 1) // ... nor more than the CALLING thread may run on right now: MODGPU_HOST_POLICY=fastest prices the loop with this number,
        if (const size_t allowed = allowed_cpus().size()) t = std::min<uint64_t>(t, allowed); // and modgpu_scalar_cycle runs it with it
    return (unsigned)std::max<uint64_t>(t, 1);
}
unsigned long long modgpu_scalar_pool_threads() { return g_pool_threads.load(); }
void modgpu_scalar_info(uint64_t out[4])
{
    out[0] = max_threads();
    out[1] = cgroup_cpu_limit();
    out[2] = allowed_cpus().size();
    out[3] = g_pool_threads.load();
}

void modgpu_scalar_cycle(uint8_t *buf, uint64_t n, int32_t key, uint64_t stream_off, int isa)
{
    const uint32_t key_res = lcg::key_residue(key);
    if (n == 0 || key_res == 0) return; // residue 0 sticks at m: keystream all zero (identity)
    const SpanFn span = kSpan[isa >= 0 && isa < MODGPU_ISA_COUNT && isa_usable(isa) ? isa : pick_isa()];
    const uint64_t pos = stream_off % lcg::PERIOD;
    uint64_t threads = modgpu_scalar_threads_for(n);
    std::vector<int> cpus;
    if (threads > 1) {
        cpus = allowed_cpus();
        if (!cpus.empty()) threads = std::min<uint64_t>(threads, cpus.size());
    }
    if (threads <= 1) {
        span(buf, n, key_res, pos);
        return;
    }
    auto call = std::make_shared<SpanCall>();
    call->span = span;
    call->buf = buf;
    call->n = n;
    call->per = ((n + threads - 1) / threads + 63) & ~63ull;
    call->count = (n + call->per - 1) / call->per;
    call->key_res = key_res;
    call->pos = pos;
    call->have_mask = sched_getaffinity(0, sizeof call->caller_mask, &call->caller_mask) == 0;
    if (spread_enabled() && cpus.size() >= call->count) {
        // span i (when a worker draws it) -> the i-th allowed CPU AFTER the one the caller is on.  Neighbours of the caller's CPU
        // are cores of its socket, next to the memory it most likely first-touched, and the same caller gets the same CPUs call
        // after call; callers on different CPUs get different neighbourhoods.  (Round 3 rotated a global start index instead, so
        // successive calls wandered over both sockets of the node: 34 GB/s at 64 MiB between 56 and 142 -- the "erratic rows".)
        const int here = sched_getcpu();
        size_t at = 0;
        for (size_t k = 0; k < cpus.size(); ++k)
            if (cpus[k] == here) at = k + 1;
        call->cpus.resize((size_t)call->count, -1);
        for (uint64_t i = 0; i < call->count; ++i) {
            if (cpus[at % cpus.size()] == here) ++at;
            call->cpus[(size_t)i] = cpus[at++ % cpus.size()];
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_pool->mu);
        const unsigned extra = (unsigned)call->count - 1;
        for (unsigned k = 0; k < extra; ++k) g_pool->requests.push_back(call);
        const int short_of = (int)g_pool->requests.size() - (int)g_pool->parked;
        for (int k = 0; k < short_of && g_pool->workers + 1 < max_threads(); ++k) {
            try {
                std::thread(pool_worker).detach();
                ++g_pool->workers;
                g_pool_threads.fetch_add(1, std::memory_order_relaxed);
            } catch (...) { // thread limit: the caller simply does more of the spans itself
                break;
            }
        }
    }
    g_pool->cv.notify_all();
    call->help(false);
    {
        std::unique_lock<std::mutex> lock(call->mu);
        call->cv.wait(lock, [&] { return call->finished == call->count; });
    }
    // entries of this call that no worker has picked up (all busy, or none could be started) are of no use to anybody now
    std::lock_guard<std::mutex> lock(g_pool->mu);
    auto &q = g_pool->requests;
    q.erase(std::remove(q.begin(), q.end(), call), q.end());
}
