// host_stream.cpp -- host-buffer and part-file endpoints of the C ABI: how bytes that start in host
// memory or in a file get through the kernel and back (include/modgpu.h: modgpu_cycle_host,
// modgpu_cycle_file*, and through them CEncryptionCycler::Cycle and the CArk part cipher).
//
// Three shapes of call (stream_impl), rates in profiles/r06_pcie_route_*.json (one MI355X behind PCIe 5 x16; ceilings of the link in
// profiles/r06_pcie_ceiling.txt: DMA 55 GB/s one way, 47 per direction both ways; one kernel in place 50.4):
//
//   in place on page-locked pages   the caller's pages are page-locked (modgpu_host_alloc): ONE kernel launch reads and writes them
//                     across PCIe where they lie -- no host copy, no DMA submissions, no device slots, no host threads.  50.2-50.4 GB/s
//                     of payload at 64 MiB .. 4 GiB, each byte crossing the link twice.  (Testing flavour, pinned mode 1: the DMA ring.)
//   one slot          <= 1 MiB, what the reference's three call sites pass (headers): one slot, one kernel across PCIe, no workers.
//   pipelined         everything else, cut into pieces for up to kPipes pipelines (host thread + two slots each) on one of five ROUTES
//                     (enum Route; each a submit / wait pair over the shared core run_route):
//                       feed         pageable memory, or a part file, that ends in memory (pageable or page-locked), below 2 GiB:
//                                    copy / pread -> pinned slot -> the call's ONE host-fed kernel (cycle_feed_kernel.h) cycles the slot
//                                    across PCIe -> copy out.  256 KiB chunks; a pipeline marks its chunk ready in page-locked memory
//                                    and polls the chunk's done word.
//                       slot_kernel  the same with a LAUNCH PER CHUNK (~64 pieces of >= 1 MiB behind a 512 KiB ramp, kernels on 4 shared
//                                    lanes): calls of 2 GiB and more, memory -> file, file -> file, a set without worker threads.
//                       in_dst       a part file of 2 GiB or more that ends in page-locked memory: pread lands in the destination, a
//                                    launch per chunk cycles it where it lies -- no slot, no copy.
//                       dma          page-locked memory that is not cycled in place (e.g. -> file): H2D DMA -> kernel in HBM -> D2H DMA.
//                     A file endpoint replaces its copy by pread / pwrite on the pinned slot.
//
// Who runs the pipelines.  A device has one staging context per NUMA node its callers' pages can be on (set 0: next to
// the GPU); a context owns a pool of slots and a pool of PARKED worker threads (started on first use, bound to the
// context's node once, never joined).  A call takes the slots it needs from the pool -- two callers on one GPU run side by
// side, each on its own slots; a call that finds too few free runs with fewer pipelines, one that finds none waits for a
// release (header-sized calls have two slots of their own) -- posts its pipelines to the workers and starts pipeline 0
// itself at once; workers and caller draw pipeline indices from the call until none is left.
//
// When the GPU is lost in the middle of a call every pipeline stops, waits for what it has in flight, and the host loop
// finishes exactly the pieces whose result has not reached the destination (Job::done, finish_on_host) -- Cycle cannot fail
// (CEncryptionCycler.cpp:4-14).  The one case that stays an error is page-locked memory cycled in place by a kernel that
// died after it was launched.  modgpu_host_trace (modgpu_testing.h) records every step of a call with a timestamp and the
// thread id; bin/modbench --route ... / --hostcall --trace print the timeline.
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <immintrin.h>
#include <time.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/modgpu_testing.h"
#include "cycle_feed_kernel.h"
#include "lcg.h"
#include "modgpu_internal.h"
#include "numa_place.h"
#include "scalar_path.h"

namespace modgpu {

// Route selectors and failure injection exist only in the testing flavour of the library
// (libmodgpu_testing.so, -DMODGPU_TESTING_HOOKS); the shipped one always takes the default routes.
#ifdef MODGPU_TESTING_HOOKS
std::atomic<int> g_pinned_mode{0};
std::atomic<int> g_staged_mode{0};
std::atomic<int> g_inject_failures{0}; // modgpu_debug_inject_failures
std::atomic<int64_t> g_inject_piece{0}; // modgpu_debug_inject_failure_at: armed while g_inject_stage >= 0, fires once
std::atomic<int> g_inject_stage{-1};
namespace {
int pinned_mode() { return g_pinned_mode.load(std::memory_order_relaxed); }
int staged_mode() { return g_staged_mode.load(std::memory_order_relaxed); }
bool injected_failure() { return g_inject_failures.load(std::memory_order_relaxed) > 0 && g_inject_failures.fetch_sub(1) > 0; }
// "the HIP call of `stage` for piece `piece` (of `pieces`) fails": true exactly once per arming
bool injected_at(uint64_t piece, uint64_t pieces, int stage)
{
    int armed = g_inject_stage.load(std::memory_order_acquire);
    if (armed != stage || pieces == 0) return false;
    const int64_t want = g_inject_piece.load(std::memory_order_relaxed);
    const uint64_t target = want == MODGPU_INJECT_PIECE_LAST ? pieces - 1 : want == MODGPU_INJECT_PIECE_MIDDLE ? pieces / 2 : std::min<uint64_t>((uint64_t)want, pieces - 1);
    return piece == target && g_inject_stage.compare_exchange_strong(armed, -1, std::memory_order_acq_rel);
}
} // namespace
#else
namespace {
constexpr int pinned_mode() { return 0; }
constexpr int staged_mode() { return 0; }
constexpr bool injected_failure() { return false; }
constexpr bool injected_at(uint64_t, uint64_t, int) { return false; }
} // namespace
#endif

namespace {

constexpr int kMaxPipes = 16;
constexpr int kSlots = 34;     // pipes x ring depth: 16 x 2 (staged) or up to 8 x 4 (direct) ...
constexpr int kPipeSlots = 32; // ... + 2 that only one-slot calls may take: a 4 KiB header Cycle never waits for a multi-GiB call to end

// ---- how a stream is cut and queued ------------------------------------------------------------------------------------------
// The shipped library reads TWO of these from the environment, once, at load:
//   MODGPU_HOST_PIPES       host threads / independent pipelines for large pageable buffers (1..16; default 8)
//   MODGPU_HOST_CHUNK_MB    largest slot in MiB (1..256; default 8)
// Everything else is a constant there -- the values the profiles named below chose -- and a knob only in the TESTING flavour
// (libmodgpu_testing.so: the same names in the environment at load, and modgpu_debug_set_host_tunable at run time), so that the
// parity suite can still drive every branch and tools/ can still sweep, but a production process has ten documented variables,
// not twenty:
//   zero_copy_max  largest buffer cycled in one pinned slot by one kernel, no chunking (1 MiB; never larger than a slot)
//   ring           device slots in flight on the DMA form of the pinned route (4)
//   split          a buffer is cut into about this many chunks ...                                   (64; 16 until round 5)
//   chunk_min      ... of at least this many bytes (and at most a slot)                               (1 MiB; 2 MiB until round 5)
//   ramp           each pipeline's FIRST and LAST chunk are this small (0 = all alike): nothing crosses the link while the first
//                  chunks are being copied in, nor while the last ones are copied out                 (512 KiB; 1 MiB until round 5)
//   lanes          streams a call's kernels-across-PCIe are queued on, in launch order, round robin (0 = every slot its own
//                  stream, the round-3 form: the GPU then runs all 16 chunk kernels of a call at once, each on a sixteenth of the
//                  link, and they all finish late together, profiles/r04_staged_midsize.txt)           (4 since round 5)
//   nt_copy        the staging copies use non-temporal stores (1)
// Chunking defaults: profiles/r05_pcie_grid.txt (16 ... 256 MiB, pageable, settings interleaved).  Round 4 had ~16 chunks of >= 2 MiB behind
// a 1 MiB ramp on 2 lanes, chosen while a chunk's kernel across the link ran at 45 % of the link's rate; with the short-launch grid of
// round 5 (modgpu_capi.cpp: kPcieGridShort) a 1-4 MiB kernel is efficient, and finer chunks -- more overlap of copy-in, link and
// copy-out -- win again: 16 / 64 / 256 MiB 34.9 / 42.5 / 45.5 GB/s with round 4's cut, 38.6 / 44.6 / 46.8 with this one.
int env_int(const char *name, int dflt, int lo, int hi)
{
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    int x = std::atoi(v);
    return x < lo ? lo : (x > hi ? hi : x);
}
#ifdef MODGPU_TESTING_HOOKS
#define MODGPU_KNOB(name, dflt, lo, hi) env_int(name, dflt, lo, hi)
#define MODGPU_KNOB_STORAGE // (changed between calls by modgpu_debug_set_host_tunable)
#else
#define MODGPU_KNOB(name, dflt, lo, hi) (dflt)
#define MODGPU_KNOB_STORAGE const
#endif
const int kPipes = env_int("MODGPU_HOST_PIPES", 8, 1, kMaxPipes);
const uint64_t kChunk = (uint64_t)env_int("MODGPU_HOST_CHUNK_MB", 8, 1, 256) << 20;
MODGPU_KNOB_STORAGE uint64_t kZeroCopyMax = std::min<uint64_t>((uint64_t)MODGPU_KNOB("MODGPU_HOST_ZEROCOPY_KB", 1024, 0, 1 << 20) << 10, kChunk);
MODGPU_KNOB_STORAGE int kRing = MODGPU_KNOB("MODGPU_HOST_RING", 4, 2, 4);
MODGPU_KNOB_STORAGE uint64_t kSplit = (uint64_t)MODGPU_KNOB("MODGPU_HOST_SPLIT", 64, 2, 256);
MODGPU_KNOB_STORAGE uint64_t kChunkMin = std::min<uint64_t>((uint64_t)MODGPU_KNOB("MODGPU_HOST_CHUNK_MIN_MB", 1, 1, 256) << 20, kChunk);
MODGPU_KNOB_STORAGE uint64_t kRamp = std::min<uint64_t>((uint64_t)MODGPU_KNOB("MODGPU_HOST_RAMP_KB", 512, 0, 1 << 18) << 10, kChunk);
MODGPU_KNOB_STORAGE int kLanes = MODGPU_KNOB("MODGPU_HOST_LANES", 4, 0, 8);
MODGPU_KNOB_STORAGE bool kNtCopy = MODGPU_KNOB("MODGPU_HOST_NTCOPY", 1, 0, 1) != 0;
constexpr int kFileLanes = 0; // calls with a file on either side keep a stream per slot: lanes made no difference there (profiles/r04_file_routes.txt)
// file_sched: which schedule FILE -> MEMORY takes (LoadArkData's part cipher): 1 (shipped) = the memory routes' (split / chunk_min / ramp /
// lanes above), 0 = the file routes' own (~16 chunks of >= 4 MiB, no ramp, a stream per slot), which memory -> file and file -> file
// always keep: their slow stage is pwrite, which wants few large calls.  Round 4 measured the memory schedule 3-5 % BEHIND on every
// file route; with round 5's short-launch grid it is ahead where the destination is memory: 33.8 / 39.7 / 44.1 -> 34.7 / 42.0 / 46.8 GB/s
// at 64 / 392 / 4096 MiB into page-locked memory, and level (within the file system's noise) the other way (profiles/r05_file_routes.txt).
MODGPU_KNOB_STORAGE int kFileSched = MODGPU_KNOB("MODGPU_HOST_FILE_SCHED", 1, 0, 1);
// feed: pageable memory on both sides is cycled by ONE host-fed kernel per call (cycle_feed_kernel.h) instead of a launch per chunk:
// uniform chunks of feed_chunk bytes (no ramp needed: nothing is launched per chunk), marked ready / done through words in page-locked
// memory.  profiles/r05_pcie_feed.txt.  0 (and staged mode 2 of the testing flavour): the launch-per-chunk schedule above.
MODGPU_KNOB_STORAGE int kFeed = MODGPU_KNOB("MODGPU_HOST_FEED", 1, 0, 1);
// file_feed: a FILE that ends in memory takes the host-fed kernel too (round 6): pread replaces the copy into the slot -- into pageable
// memory and, below 2 GiB, into page-locked memory as well (three ways were measured for that destination, profiles/r06_file_routes.txt:
// through the slots 43.5 / 47.0 / 47.3 GB/s at 64 / 392 / 1024 MiB, one host-fed kernel working in the destination itself 42.3 / 44.2 /
// 44.4, round 5's launch per chunk in the destination 38.8 / 43.8 / 46.1).  0 (testing flavour): round 5's launch per chunk.
MODGPU_KNOB_STORAGE int kFileFeed = MODGPU_KNOB("MODGPU_HOST_FILE_FEED", 1, 0, 1);
MODGPU_KNOB_STORAGE uint64_t kFeedChunk = (uint64_t)MODGPU_KNOB("MODGPU_HOST_FEED_CHUNK_KB", 256, 32, 8192) << 10;
constexpr uint32_t kFeedChunksMax = 8192;      // ready / done words per call (larger calls take larger chunks)
constexpr uint64_t kFeedBelow = 2ull << 30;    // from here up a launch per 8 MiB chunk is as good or better (4 GiB, pageable: 49.3 against 48.7 GB/s, profiles/r05_pcie_feed.txt;
                                               // file -> pageable 46.4 / 47.0, profiles/r06_file_routes.txt)
constexpr uint32_t kFeedGrid = 32;             // workgroups of the host-fed kernel: what saturates the link (profiles/r05_pcie_persist.txt)
MODGPU_KNOB_STORAGE uint64_t kFeedPatienceTicks = 1000000000ull; // 10 s of the 100 MHz wall clock: a chunk the host has not delivered by then never comes

// ---- host-side timeline of the staged / pinned routes (modgpu_host_trace, reporting only) --------------------------------
std::atomic<bool> g_trace_on{false};
std::mutex g_trace_mu;
std::vector<modgpu_host_trace_event_t> g_trace;
inline void trace(int kind, int pipe, uint64_t chunk, uint64_t bytes)
{
    if (!g_trace_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lock(g_trace_mu);
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts); // (read under the lock: the log is in time order)
    static thread_local const int tid = (int)::syscall(SYS_gettid); // (what rocprofv3's kernel trace calls Thread_Id)
    if (g_trace.size() < (1u << 20)) g_trace.push_back({(uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec, kind, pipe, chunk, bytes, tid, 0});
}

struct Call;
// One per logical device, allocated once and never destroyed (parked workers wait on its condition variable for the life of
// the process).  `mu` guards the slot ownership table and the work queue; a slot's buffers and stream are touched only by the
// call that owns the slot.
struct Staging {
    std::mutex mu;
    std::condition_variable slot_cv; // a call gave slots back
    std::condition_variable work_cv; // a call posted pipelines
    uint8_t *pinned[kSlots] = {};
    uint8_t *dev[kSlots] = {};
    hipStream_t stream[kSlots] = {};
    uint64_t pinned_cap[kSlots] = {};
    uint64_t dev_cap[kSlots] = {};
    hipEvent_t event[kSlots] = {};              // "the kernel queued for this slot has finished" when it ran on a shared lane
    bool busy[kSlots] = {};                     // under mu
    std::deque<std::shared_ptr<Call>> requests; // under mu: one entry per pipeline a call would like a worker for
    int workers = 0, parked = 0;                // under mu
    std::atomic<int> node{-1}; // NUMA node this set's slots and workers live on; -1: next to the GPU (where the runtime puts page-locked memory)
    bool placed[kSlots] = {}; // the slot's pinned buffer is our own placed mapping (numa::reserve + hipHostRegister), not hipHostMalloc's
    // what a host-fed call needs besides its slots, kept with the call's FIRST slot: page-locked host-coherent words
    // [0, kFeedChunksMax) ready, [kFeedChunksMax, 2 kFeedChunksMax) done, [2 kFeedChunksMax] abort -- and the kernel's device words
    uint32_t *feed_flags[kSlots] = {};
    uint32_t *feed_flags_dev[kSlots] = {};
    uint32_t *feed_work[kSlots] = {};
};
// A device has one staging context per NUMA node a caller's pages can be on (round 5, profiles/r05_staged_numa.txt).  The staged
// route is two CPU copies of every byte, and a copy whose SOURCE is on the other socket runs at 6-12 GB/s per thread instead of 28:
// with the slots next to the GPU (where hipHostMalloc puts them) and the caller's pages on the other socket, either the copy in or
// the copy out reads remotely whichever socket the workers run on -- pageable 64 MiB fell from 44 to 20-35 GB/s, in about half of
// all processes on a two-socket node.  So a call whose pages live on another node than the GPU takes THAT node's set: slots
// allocated there, workers bound there; both copies stay inside one socket and only the GPU crosses to the other -- which costs
// it nothing (one kernel across PCIe on the other socket's memory runs at the same 50 GB/s, profiles/r03_numa.txt).
constexpr int kNodeSets = 9; // set 0: next to the GPU (also: node unknown, file endpoints, nodes beyond 7); sets 1..8: NUMA nodes 0..7 (two sockets in NPS4)
// Contexts are made on first use (most processes use one or two of the 576) and never destroyed.
std::atomic<Staging *> g_staging[kMaxDevices * kNodeSets];
Staging &staging_of(int dev, int set)
{
    std::atomic<Staging *> &at = g_staging[dev * kNodeSets + set];
    Staging *s = at.load(std::memory_order_acquire);
    if (!s) {
        Staging *fresh = new Staging;
        if (at.compare_exchange_strong(s, fresh, std::memory_order_acq_rel)) s = fresh;
        else delete fresh; // another caller was first
    }
    return *s;
}
// fork(): the child has neither the parked workers nor a usable HIP context; it starts with no staging contexts at all (the
// old ones, whose mutexes a vanished thread may hold, are leaked on purpose).  ADVICE r4.
const int g_staging_atfork = ::pthread_atfork(nullptr, nullptr, [] {
    for (auto &p : g_staging) p.store(nullptr, std::memory_order_relaxed);
});
std::atomic<uint64_t> g_pool_spawned{0}, g_pool_tasks{0}, g_slot_waits{0}, g_calls_in_flight{0}, g_calls_overlapped{0}, g_node_set_calls{0};

// A device whose host-fed kernel stopped responding (see feed_stop): its staging sets take no further call -- whatever is queued behind
// a kernel that never ends would never run --, the slots the lost call held are never handed out again, and Cycle serves the device's
// callers from the host loop from then on (the attempt fails before it touches anything: modgpu_cycle_auto_host's second branch).
std::atomic<bool> g_device_lost[kMaxDevices];
int fail_lost() { return fail(MODGPU_ERR_HIP, "this device's host-buffer routes were abandoned: a host-fed kernel stopped responding"); }

// Slots a call owns, given back (and waiters woken) when the call ends, whichever way.
struct SlotLease {
    Staging &s;
    const int dev;
    std::vector<int> ids; // (empty after acquire: the device was lost while this call waited)
    SlotLease(Staging &st, int dev_) : s(st), dev(dev_) {}
    SlotLease(const SlotLease &) = delete;
    SlotLease &operator=(const SlotLease &) = delete;
    // takes up to `want` slots in whole groups of `group`; waits while fewer than one group is free.  A call that asks for ONE
    // slot (header-sized buffers, page-locked memory cycled in place) may take any, the last two first; every other call only the
    // first kPipeSlots -- so the one-slot routes find a slot however many large calls are at work.
    void acquire(int want, int group)
    {
        std::unique_lock<std::mutex> lock(s.mu);
        const bool single = want == 1 && group == 1;
        const int limit = single ? kSlots : kPipeSlots;
        auto n_free = [&] {
            int f = 0;
            for (int i = 0; i < limit; ++i) f += s.busy[i] ? 0 : 1;
            return f;
        };
        auto lost = [&] { return g_device_lost[dev].load(std::memory_order_acquire); };
        if (n_free() < group && !lost()) {
            g_slot_waits.fetch_add(1, std::memory_order_relaxed);
            s.slot_cv.wait(lock, [&] { return n_free() >= group || lost(); });
        }
        if (lost()) return;
        const int take = std::min(want, n_free()) / group * group;
        for (int k = 0; k < limit && (int)ids.size() < take; ++k) {
            const int i = single ? limit - 1 - k : k;
            if (!s.busy[i]) {
                s.busy[i] = true;
                ids.push_back(i);
            }
        }
    }
    ~SlotLease()
    {
        if (ids.empty() || g_device_lost[dev].load(std::memory_order_acquire)) return; // (a lost device's slots stay taken: a kernel may still be at work in them)
        {
            std::lock_guard<std::mutex> lock(s.mu);
            for (int i : ids) s.busy[i] = false;
        }
        s.slot_cv.notify_all();
    }
};

// The slots of a lease get a stream, `need` device bytes and -- if want_pinned -- `need` pinned bytes each (grown on
// demand, never shrunk).  `need` is clamped to [1 MiB, kChunk].  Only the owner of a slot touches it: no lock.
int staging_reserve(Staging &s, const std::vector<int> &ids, uint64_t need, bool want_dev, bool want_pinned)
{
    need = std::min<uint64_t>(std::max<uint64_t>(need, 1ull << 20), kChunk);
    for (int i : ids) {
        if (!s.stream[i]) HIP_TRY(hipStreamCreateWithFlags(&s.stream[i], hipStreamNonBlocking));
        if (!s.event[i]) HIP_TRY(hipEventCreateWithFlags(&s.event[i], hipEventDisableTiming));
        if (want_pinned && s.pinned_cap[i] < need) {
            if (s.pinned[i]) {
                if (s.placed[i]) {
                    HIP_TRY(hipHostUnregister(s.pinned[i]));
                    numa::release(s.pinned[i], s.pinned_cap[i]);
                } else HIP_TRY(hipHostFree(s.pinned[i]));
            }
            s.pinned[i] = nullptr;
            s.pinned_cap[i] = 0;
            s.placed[i] = false;
            if (const int node = s.node.load(std::memory_order_relaxed); node >= 0) { // this set's slots live on a named node: our own mapping, bound, touched, then page-locked in place
                void *p = numa::reserve(need);
                if (p && numa::prefer_node(p, need, node) == 0) {
                    numa::prefault(p, need, 1, "/sys", -1);
                    if (hipHostRegister(p, need, hipHostRegisterPortable | hipHostRegisterMapped) == hipSuccess) {
                        s.pinned[i] = static_cast<uint8_t *>(p);
                        s.placed[i] = true;
                        p = nullptr;
                    } else (void)hipGetLastError();
                }
                if (p) numa::release(p, need); // (no placement to be had: the runtime's own allocation below)
            }
            if (!s.pinned[i]) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s.pinned[i]), need, hipHostMallocPortable | hipHostMallocMapped));
            s.pinned_cap[i] = need;
        }
        if (want_dev && s.dev_cap[i] < need) {
            if (s.dev[i]) HIP_TRY(hipFree(s.dev[i]));
            s.dev[i] = nullptr;
            s.dev_cap[i] = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s.dev[i]), need));
            s.dev_cap[i] = need;
        }
    }
    return MODGPU_OK;
}

// The flag words and device counters of a host-fed call (made once per slot that ever leads such a call).
int feed_reserve(Staging &s, int slot)
{
    if (!s.feed_flags[slot]) {
        uint32_t *h = nullptr;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h), (2 * kFeedChunksMax + 16) * sizeof(uint32_t), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&s.feed_flags_dev[slot]), h, 0) != hipSuccess) {
            (void)hipHostFree(h);
            return fail(MODGPU_ERR_HIP, "hipHostGetDevicePointer (host-fed kernel's flag words)");
        }
        s.feed_flags[slot] = h;
    }
    if (!s.feed_work[slot]) { // zero when a call finds them: cleared here once, and by every call behind itself
        // Cleared ON THE STREAM THE KERNEL WILL BE LAUNCHED ON (the slot's own; staging_reserve has made it).  A plain hipMemset goes to the
        // NULL stream, which this non-blocking stream does not wait for, and it may return before the fill has run: round 6 starts the
        // worker threads before this point, so the launch now follows within microseconds -- and with eight devices' first calls at once
        // the kernel ran while the fill was still queued: the ticket counter went back to zero under it, pieces were drawn twice and
        // chunks marked done early (13 of 25 fresh processes returned wrong bytes; gpurun_out/r06e_diag_many.log).
        uint32_t *w = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w), (kFeedChunksMax + 2) * sizeof(uint32_t)));
        if (hipMemsetAsync(w, 0, (kFeedChunksMax + 2) * sizeof(uint32_t), s.stream[slot]) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(w);
            return fail(MODGPU_ERR_HIP, "hipMemsetAsync (host-fed kernel's counters)");
        }
        s.feed_work[slot] = w;
    }
    return MODGPU_OK;
}

// ---- the staging copies ------------------------------------------------------------------------------------------------------
// What bounds the staged route since round 4 is its two CPU copies of every byte (profiles/r04_staged_midsize.txt): ~10 GB/s per
// thread, because both are DRAM-miss streams -- and a plain memcpy of a few MiB also READS every destination line before it
// overwrites it (write-allocate).  A copy with non-temporal stores does not: two DRAM streams instead of three.  The bytes are
// not wanted in this core's cache anyway -- the slot is read next by the GPU across PCIe, the caller's buffer by whoever comes next.
// (nt_copy = 0, testing flavour: plain memcpy.)
__attribute__((target("avx2"))) void copy_nt_avx2(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    // head: up to the first 32-byte boundary of dst
    const uint64_t head = std::min<uint64_t>(n, (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31);
    std::memcpy(dst, src, head);
    dst += head, src += head, n -= head;
    uint64_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 32)),
                      c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 64)), d = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(dst + i + 96), d);
    }
    _mm_sfence(); // the streamed lines are globally visible before the kernel is launched / the call returns
    std::memcpy(dst + i, src + i, n - i);
}
void copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (kNtCopy && avx2 && n >= (256u << 10)) copy_nt_avx2(dst, src, n);
    else std::memcpy(dst, src, n);
}

// `len` bytes at offset `off` of the source -> to[0 .. len)
int fill_slot(const Endpoint &src, uint8_t *to, uint64_t off, uint64_t len)
{
    if (src.mem) {
        copy_bytes(to, src.mem + off, len);
        return MODGPU_OK;
    }
    for (uint64_t done = 0; done < len;) {
        ssize_t r = ::pread(src.fd, to + done, len - done, (off_t)(src.base + off + done));
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return fail_io("pread");
        if (r == 0) return fail(MODGPU_ERR_IO, "pread: unexpected end of file");
        done += (uint64_t)r;
    }
    return MODGPU_OK;
}

int drain_slot(const Endpoint &dst, const uint8_t *pinned, uint64_t off, uint64_t len)
{
    if (dst.mem) {
        copy_bytes(dst.mem + off, pinned, len);
        return MODGPU_OK;
    }
    for (uint64_t done = 0; done < len;) {
        ssize_t r = ::pwrite(dst.fd, pinned + done, len - done, (off_t)(dst.base + off + done));
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return fail_io("pwrite");
        done += (uint64_t)r;
    }
    return MODGPU_OK;
}

// How a call's chunks get through the GPU.  One route per call; each is a pair of functions below (submit / wait) over the
// shared pipeline core (run_route): the ring of slots, the order of steps, the failure rule and the copy out are the same for all.
enum class Route {
    dma,          // H2D DMA -> kernel in HBM -> D2H DMA, through device slots (page-locked caller memory on a side that is not cycled in
                  // place; testing flavour: staged mode 1)
    slot_kernel,  // copy / pread into a pinned slot -> ONE LAUNCH PER CHUNK across PCIe on the slot -> copy / pwrite out
    in_dst,       // file -> page-locked caller memory: pread lands in the destination itself, a launch per chunk cycles it where it lies
    feed,         // as slot_kernel, but ONE host-fed kernel for the whole call (cycle_feed_kernel.h): a pipeline marks its chunk ready
                  // in page-locked memory and polls the chunk's done word
};

// What a call knows about each piece of its stream.  The cipher is positional and the pieces are disjoint, so a call that loses
// its GPU half-way can be FINISHED by the host loop over exactly the pieces whose result has not reached the destination
// (the reference's Cycle cannot fail, CEncryptionCycler.cpp:4-14, and its callers do not guard it).
struct Job {
    const Endpoint &src, &dst;
    uint64_t n, chunk; // chunk: the largest piece (slot size)
    int32_t key;
    uint64_t stream_off;
    Route route = Route::dma;
    std::atomic<bool> touched{false}; // the destination may differ from what it was
    std::atomic<bool> failed{false};  // a pipeline failed: the others stop filling and launching at once
    std::vector<Piece> plan;          // the stream cut into pieces, in stream order; piece k belongs to pipeline k mod pipes
    std::unique_ptr<std::atomic<uint8_t>[]> done; // per piece: its result is in the destination, whole
    std::vector<hipStream_t> lanes; // kernels across PCIe are queued on these in launch order (empty: each on its slot's stream)
    std::atomic<uint64_t> launched{0};
    // host-fed call (cycle_feed_kernel.h): one kernel for the whole call; a pipeline marks a chunk ready instead of launching, and polls its done word
    uint32_t *feed_ready = nullptr, *feed_done = nullptr, *feed_abort = nullptr; // host addresses
    hipStream_t feed_stream = nullptr;
    std::atomic<bool> feed_launched{false}; // the kernel is on feed_stream (it is launched while the pipelines copy their first chunks in)
    int dev = 0;                            // logical device (whose host-buffer routes are abandoned if the kernel stops responding)
    std::atomic<int64_t> feed_progress_ns{0}; // steady-clock time at which a pipeline last saw one of the call's chunks done: the kernel is alive
    int copy_node = -1; // NUMA node the caller's pages live on (-1: unknown, or no pageable memory endpoint): picks the staging set (g_staging)
    cpu_set_t caller_mask; // the calling thread's affinity mask: a worker is never put on a CPU the caller may not use
    bool have_mask = false;
    Job(const Endpoint &s, const Endpoint &d, uint64_t n_, uint64_t chunk_, int32_t key_, uint64_t off_) : src(s), dst(d), n(n_), chunk(chunk_), key(key_), stream_off(off_) {}
    bool fed() const { return route == Route::feed; }
    // the device writes the caller's destination itself (DMA or a kernel in place): an unfinished piece of it is undefined
    bool dst_written_by_device() const { return route == Route::in_dst || (route == Route::dma && dst.mem && dst.pinned); }
    void set_plan(std::vector<Piece> p)
    {
        plan = std::move(p);
        done.reset(new std::atomic<uint8_t>[plan.size() ? plan.size() : 1]);
        for (size_t k = 0; k < plan.size(); ++k) done[k].store(0, std::memory_order_relaxed);
    }
};

// Cuts [0, n) into pieces of `chunk` bytes for `pipes` pipelines.  With a ramp the first and the last `pipes` pieces -- every
// pipeline's first and last -- are only `ramp` bytes: the link carries nothing while the first pieces are copied into their slots
// and nothing while the last are copied out, and that exposed time shrinks with them (profiles/r04_staged_midsize.txt).
std::vector<Piece> cut_stream(uint64_t n, uint64_t chunk, int pipes, uint64_t ramp)
{
    std::vector<Piece> plan;
    uint64_t at = 0;
    const uint64_t edge = (uint64_t)pipes * ramp;
    const bool ramped = ramp > 0 && ramp < chunk && pipes > 1 && n >= 2 * edge + (uint64_t)pipes * chunk;
    if (ramped)
        for (int p = 0; p < pipes; ++p, at += ramp) plan.push_back({at, ramp});
    const uint64_t middle_end = ramped ? n - edge : n;
    for (; at < middle_end; at += chunk) plan.push_back({at, std::min<uint64_t>(chunk, middle_end - at)});
    if (ramped)
        for (at = middle_end; at < n; at += ramp) plan.push_back({at, std::min<uint64_t>(ramp, n - at)});
    return plan;
}

constexpr int kStopped = -1000; // run_pipe: another pipeline of the call failed and this one stopped early -- not an error of its own

// How long the host puts up with a host-fed kernel that neither finishes a chunk nor ends.  The kernel itself gives a chunk up after
// its patience P and is gone microseconds later, so a healthy call never waits anywhere near 4 P + grace (50 s by default) for one chunk;
// a kernel that is still "running" by then has stopped responding (a workgroup wedged at a barrier: what check_isa.py's EXEC rule keeps
// out of a build, and what a host must survive anyway).
std::chrono::duration<double> feed_grace()
{
    const double p = (double)kFeedPatienceTicks * 1e-8;
    return std::chrono::duration<double>(std::min(10.0, std::max(1.0, 10.0 * p)));
}
std::chrono::duration<double> feed_host_deadline() { return std::chrono::duration<double>(4.0 * (double)kFeedPatienceTicks * 1e-8) + feed_grace(); }

// Waits until the host-fed kernel has marked chunk c done.  Spins (the wait is tens of microseconds while the call is healthy); every
// ~50 us it looks at the call -- a sibling pipeline that failed --, every millisecond at the kernel's stream -- a kernel that has ended
// without finishing the chunk gave up (it waited too long for the host) or died, and either way the chunk will never be done by it --, and
// from then on it gives the CPU away between looks (a host with fewer CPUs than pipelines must not burn its quota here).  A kernel that
// neither ends nor has finished ANY chunk of the call for feed_host_deadline() has stopped responding (a slow kernel that keeps finishing
// chunks -- the sanitizers' stand-in slowed ten times -- has not): the call is lost like any other.
int64_t steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int feed_wait(Job &j, uint64_t c)
{
    uint32_t spins = 0;
    const int64_t began = steady_ns();
    struct Progress { // (whichever way the wait ends with the chunk done, the kernel was alive just now)
        Job &j;
        const uint64_t c;
        ~Progress() { if (__atomic_load_n(&j.feed_done[c], __ATOMIC_ACQUIRE) != 0u) j.feed_progress_ns.store(steady_ns(), std::memory_order_relaxed); }
    } progress{j, c};
    while (__atomic_load_n(&j.feed_done[c], __ATOMIC_ACQUIRE) == 0u) {
        _mm_pause();
        if ((++spins & 1023u) != 0) continue;
        if (j.failed.load(std::memory_order_acquire)) return kStopped;
        if ((spins & 16383u) != 0 || !j.feed_launched.load(std::memory_order_acquire)) continue; // (an empty stream is idle, too)
        const hipError_t q = hipStreamQuery(j.feed_stream);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            const int64_t quiet_since = std::max(began, j.feed_progress_ns.load(std::memory_order_relaxed));
            if ((double)(steady_ns() - quiet_since) * 1e-9 > feed_host_deadline().count())
                return fail(MODGPU_ERR_HIP, "the host-fed kernel stopped responding (it neither finished a chunk nor ended)");
            std::this_thread::yield();
            continue;
        }
        if (__atomic_load_n(&j.feed_done[c], __ATOMIC_ACQUIRE) != 0u) break; // (it finished the chunk and then ended)
        return q == hipSuccess ? fail(MODGPU_ERR_HIP, "the host-fed kernel ended before the chunk was done (it gave up waiting for the host)")
                               : fail_hip(q, "hipStreamQuery (host-fed kernel)");
    }
    return MODGPU_OK;
}

// Makes sure the call's host-fed kernel is gone: raises abort (a healthy kernel leaves at its next look, microseconds to milliseconds) and
// waits for its stream to go idle -- but not for ever.  A kernel still there after the grace period will not leave: the device's
// host-buffer routes are abandoned (g_device_lost) instead of this thread hanging in hipStreamSynchronize with it.  Returns false then.
bool feed_stop(Job &j)
{
    __atomic_store_n(j.feed_abort, 1u, __ATOMIC_RELEASE);
    if (!j.feed_launched.load(std::memory_order_acquire)) return true;
    const auto began = std::chrono::steady_clock::now();
    const auto until = began + feed_grace();
    for (;;) {
        if (g_device_lost[j.dev].load(std::memory_order_acquire)) return false; // (a sibling pipeline has found out already)
        const hipError_t q = hipStreamQuery(j.feed_stream);
        if (q != hipErrorNotReady) {
            (void)hipGetLastError();
            return true;
        }
        (void)hipGetLastError();
        const auto now = std::chrono::steady_clock::now();
        if (now > until) break;
        if (now - began < std::chrono::microseconds(300)) // (the normal case: the kernel is on its way out -- a few microseconds)
            for (int k = 0; k < 64; ++k) _mm_pause();
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    g_device_lost[j.dev].store(true, std::memory_order_release);
    trace(MODGPU_TRACE_FAILED, -1, 0, 99);
    return false;
}

// ---- one pipeline -----------------------------------------------------------------------------------------------------------------
// Chunks first, first + stride, ... of the stream through the `ring` slots slots[0..ring).
// Failure: the pipeline that meets it raises j.failed, every pipeline sees that at its next step and stops; each waits for what
// it has in flight and returns.  Pieces whose result had reached the destination are marked in j.done.  THE INVARIANT the mid-call
// rescue rests on: on the routes that pass through a slot (dma with a pageable destination, slot_kernel, feed) a piece changes
// the caller's destination in exactly one place -- drain_slot in retire() below, after the piece's wait has succeeded -- and
// j.touched is raised right in front of it; the routes on which the DEVICE writes the destination (dma into page-locked memory,
// in_dst: Job::dst_written_by_device) raise it in their submit.
struct Pipe {
    Staging &s;
    const int *slots;
    int ring;
    Job &j;
    uint64_t first, stride;
    int pipe;
    uint64_t n_chunks, mine;
    bool src_direct, dst_direct, on_lanes;
    uint64_t chunk_of(uint64_t i) const { return first + i * stride; }
    // a kernel that works across PCIe (on the slot, or on the caller's page-locked destination): on the call's shared lanes, if it has any
    int launch_across_pcie(void *mapped, uint64_t len, uint64_t off, int slot)
    {
        hipStream_t st = on_lanes ? j.lanes[j.launched.fetch_add(1, std::memory_order_relaxed) % j.lanes.size()] : s.stream[slot];
        int rc = cycle_device_impl(mapped, len, j.key, j.stream_off + off, st, /*over_pcie=*/true);
        if (rc == MODGPU_OK && on_lanes) HIP_TRY(hipEventRecord(s.event[slot], st));
        return rc;
    }
    // (testing flavour) "the HIP call of `stage` for piece c fails": true once per arming; the caller returns the error
    bool inject(uint64_t c, int stage)
    {
        if (!injected_at(c, n_chunks, stage)) return false;
        j.failed.store(true, std::memory_order_release); // (before the trace line: what follows it in the log has seen it)
        trace(MODGPU_TRACE_FAILED, pipe, c, (uint64_t)stage);
        (void)fail(MODGPU_ERR_HIP, "injected failure (modgpu_debug_inject_failure_at)");
        return true;
    }
};
#define MODGPU_INJECT(p, piece, stage)                \
    do {                                              \
        if ((p).inject((piece), (stage))) return MODGPU_ERR_HIP; \
    } while (0)

// -- the routes: submit(p, c, slot) gets chunk c from the source to where the kernel works on it and has the kernel started (or
//    told); wait(p, c, slot) returns once the chunk's result is whole -- in the slot, or in the destination on the direct routes.
struct RouteDma {
    static bool drains(const Pipe &p) { return !p.dst_direct; }
    static int submit(Pipe &p, uint64_t c, int slot)
    {
        Job &j = p.j;
        Staging &s = p.s;
        const uint64_t off = j.plan[c].off, len = j.plan[c].len;
        if (p.src_direct) {
            HIP_TRY(hipMemcpyAsync(s.dev[slot], j.src.mem + off, len, hipMemcpyHostToDevice, s.stream[slot]));
        } else {
            int rc = fill_slot(j.src, s.pinned[slot], off, len);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(s.dev[slot], s.pinned[slot], len, hipMemcpyHostToDevice, s.stream[slot]));
        }
        trace(MODGPU_TRACE_FILL_END, p.pipe, c, len);
        MODGPU_INJECT(p, c, MODGPU_STAGE_LAUNCH);
        int rc = cycle_device_impl(s.dev[slot], len, j.key, j.stream_off + off, s.stream[slot]);
        if (rc) return rc;
        trace(MODGPU_TRACE_LAUNCHED, p.pipe, c, len);
        if (p.dst_direct) {
            j.touched.store(true, std::memory_order_relaxed);
            HIP_TRY(hipMemcpyAsync(j.dst.mem + off, s.dev[slot], len, hipMemcpyDeviceToHost, s.stream[slot]));
        } else {
            HIP_TRY(hipMemcpyAsync(s.pinned[slot], s.dev[slot], len, hipMemcpyDeviceToHost, s.stream[slot]));
        }
        return MODGPU_OK;
    }
    static int wait(Pipe &p, uint64_t, int slot)
    {
        HIP_TRY(hipStreamSynchronize(p.s.stream[slot]));
        return MODGPU_OK;
    }
};
struct RouteSlotKernel { // neither side is page-locked caller memory: the slot itself is the device-visible copy
    static bool drains(const Pipe &) { return true; }
    static int submit(Pipe &p, uint64_t c, int slot)
    {
        Job &j = p.j;
        const uint64_t off = j.plan[c].off, len = j.plan[c].len;
        int rc = fill_slot(j.src, p.s.pinned[slot], off, len);
        if (rc) return rc;
        trace(MODGPU_TRACE_FILL_END, p.pipe, c, len);
        MODGPU_INJECT(p, c, MODGPU_STAGE_LAUNCH);
        void *mapped = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&mapped, p.s.pinned[slot], 0));
        rc = p.launch_across_pcie(mapped, len, off, slot);
        trace(MODGPU_TRACE_LAUNCHED, p.pipe, c, len);
        return rc;
    }
    static int wait(Pipe &p, uint64_t, int slot)
    {
        if (p.on_lanes) HIP_TRY(hipEventSynchronize(p.s.event[slot]));
        else HIP_TRY(hipStreamSynchronize(p.s.stream[slot]));
        return MODGPU_OK;
    }
};
struct RouteInDst {
    static bool drains(const Pipe &) { return false; }
    static int submit(Pipe &p, uint64_t c, int slot)
    {
        Job &j = p.j;
        const uint64_t off = j.plan[c].off, len = j.plan[c].len;
        j.touched.store(true, std::memory_order_relaxed);
        int rc = fill_slot(j.src, j.dst.mem + off, off, len);
        if (rc) return rc;
        trace(MODGPU_TRACE_FILL_END, p.pipe, c, len);
        MODGPU_INJECT(p, c, MODGPU_STAGE_LAUNCH);
        void *mapped = nullptr;
        HIP_TRY(hipHostGetDevicePointer(&mapped, j.dst.mem + off, 0));
        rc = p.launch_across_pcie(mapped, len, off, slot);
        trace(MODGPU_TRACE_LAUNCHED, p.pipe, c, len);
        return rc;
    }
    static int wait(Pipe &p, uint64_t c, int slot) { return RouteSlotKernel::wait(p, c, slot); }
};
// (testing flavour) the host goes away: the pipeline holds its chunk back UNTIL THE KERNEL HAS GIVEN THE CALL UP -- its stream has gone
// idle with the chunk undone --, not for a time it guesses: who arrives at the chunk first, the kernel or the pipeline, is the box's
// business (under TSan the stand-in kernel needs > 100 ms to get there), and the case is "the kernel waits and nobody comes" either
// way.  Bounded, so that a kernel that never leaves shows up as the call's own failure further down, not as a hung test.
void stall_until_the_kernel_has_left(Pipe &p, uint64_t c)
{
    Job &j = p.j;
    if (!j.fed() || !injected_at(c, p.n_chunks, MODGPU_STAGE_STALL)) return;
    trace(MODGPU_TRACE_FAILED, p.pipe, c, (uint64_t)MODGPU_STAGE_STALL);
    const auto until = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    while (std::chrono::steady_clock::now() < until && !j.failed.load(std::memory_order_acquire)) {
        if (j.feed_launched.load(std::memory_order_acquire)) {
            if (hipStreamQuery(j.feed_stream) != hipErrorNotReady) break;
            (void)hipGetLastError();
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
}
struct RouteFeed { // the kernel is there already, waiting for exactly this chunk's ready word
    static bool drains(const Pipe &) { return true; }
    static int submit(Pipe &p, uint64_t c, int slot)
    {
        Job &j = p.j;
        const uint64_t off = j.plan[c].off, len = j.plan[c].len;
        stall_until_the_kernel_has_left(p, c);
        int rc = fill_slot(j.src, p.s.pinned[slot], off, len);
        if (rc) return rc;
        trace(MODGPU_TRACE_FILL_END, p.pipe, c, len);
        MODGPU_INJECT(p, c, MODGPU_STAGE_LAUNCH);
        __atomic_store_n(&j.feed_ready[c], 1u, __ATOMIC_RELEASE);
        trace(MODGPU_TRACE_READY, p.pipe, c, len);
        return MODGPU_OK;
    }
    static int wait(Pipe &p, uint64_t c, int) { return feed_wait(p.j, c); }
};
// -- the core every route runs on: step i retires the chunk that used this step's slot `ring` steps ago, then submits chunk i
template <class R> int retire(Pipe &p, uint64_t c, int slot)
{
    Job &j = p.j;
    const uint64_t off = j.plan[c].off, len = j.plan[c].len;
    trace(MODGPU_TRACE_SYNC_BEGIN, p.pipe, c, len);
    MODGPU_INJECT(p, c, MODGPU_STAGE_SYNC);
    int rc = R::wait(p, c, slot);
    if (rc) return rc;
    trace(MODGPU_TRACE_SYNC_END, p.pipe, c, len);
    if (R::drains(p)) {
        MODGPU_INJECT(p, c, MODGPU_STAGE_DRAIN);
        j.touched.store(true, std::memory_order_relaxed);
        rc = drain_slot(j.dst, p.s.pinned[slot], off, len); // THE place a slot route writes the caller's destination
        trace(MODGPU_TRACE_DRAIN_END, p.pipe, c, len);
        if (rc) return rc;
    }
    j.done[c].store(1, std::memory_order_release); // (a destination written directly -- DMA, or the kernel in place -- holds the piece once the wait has succeeded)
    MODGPU_INJECT(p, c, MODGPU_STAGE_AFTER_DRAIN);
    return MODGPU_OK;
}
template <class R> int run_route(Pipe &p)
{
    Job &j = p.j;
    for (uint64_t i = 0; i < p.mine + (uint64_t)p.ring; ++i) {
        const int slot = p.slots[i % (uint64_t)p.ring];
        if (j.failed.load(std::memory_order_acquire)) return kStopped;
        if (i >= (uint64_t)p.ring) {
            const int rc = retire<R>(p, p.chunk_of(i - (uint64_t)p.ring), slot);
            if (rc) return rc;
            if (j.failed.load(std::memory_order_acquire)) return kStopped;
        }
        if (i < p.mine) {
            const uint64_t c = p.chunk_of(i);
            trace(MODGPU_TRACE_FILL_BEGIN, p.pipe, c, j.plan[c].len);
            MODGPU_INJECT(p, c, MODGPU_STAGE_FILL);
            const int rc = R::submit(p, c, slot);
            if (rc) return rc;
        }
    }
    return MODGPU_OK;
}
#undef MODGPU_INJECT

int run_pipe(Staging &s, const int *slots, int ring, Job &j, uint64_t first, uint64_t stride)
{
    const uint64_t n_chunks = j.plan.size();
    Pipe p{s, slots, ring, j, first, stride, (int)first, n_chunks, first < n_chunks ? (n_chunks - first + stride - 1) / stride : 0,
           j.src.mem && j.src.pinned, j.dst.mem && j.dst.pinned,
           !j.lanes.empty() && (j.route == Route::slot_kernel || j.route == Route::in_dst)};
    trace(MODGPU_TRACE_PIPE_START, p.pipe, 0, 0);
    int rc = MODGPU_OK;
    switch (j.route) {
    case Route::dma: rc = run_route<RouteDma>(p); break;
    case Route::slot_kernel: rc = run_route<RouteSlotKernel>(p); break;
    case Route::in_dst: rc = run_route<RouteInDst>(p); break;
    case Route::feed: rc = run_route<RouteFeed>(p); break;
    }
    if (rc != MODGPU_OK) { // nothing of this call may still be running against the caller's memory (or its slots) once we return
        if (rc != kStopped) j.failed.store(true, std::memory_order_release);
        const std::string keep = t_err;
        if (j.fed()) (void)feed_stop(j); // the kernel leaves at its next look at the flag; chunks it has not finished stay undone
        for (int k = 0; k < ring; ++k)
            if (!j.fed() || s.stream[slots[k]] != j.feed_stream) (void)hipStreamSynchronize(s.stream[slots[k]]); // (never wait unboundedly on the kernel's own stream)
        for (hipStream_t st : j.lanes) (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        t_err = keep;
    }
    trace(MODGPU_TRACE_PIPE_END, p.pipe, 0, 0);
    return rc;
}

// ---- a call's pipelines, drawn by the caller and by parked workers ---------------------------------------------------------
struct Call {
    Staging &s;
    Job &job; // (lives on the caller's stack: touched only while a pipeline index is held, and the caller waits for those)
    const int pipes, ring, phys;
    std::vector<int> slots; // pipes * ring slot ids
    std::atomic<int> next{0};
    std::mutex mu;
    std::condition_variable cv;
    int finished = 0; // under mu
    std::vector<int> rcs;
    std::vector<std::string> errs;
    Call(Staging &st, Job &j, int p, int r, int ph, std::vector<int> ids) : s(st), job(j), pipes(p), ring(r), phys(ph), slots(std::move(ids)), rcs((size_t)p, MODGPU_OK), errs((size_t)p) {}
    // runs pipelines until none is left to start
    void help(bool worker)
    {
        for (;;) {
            const int p = next.fetch_add(1, std::memory_order_relaxed);
            if (p >= pipes) return;
            if (worker) g_pool_tasks.fetch_add(1, std::memory_order_relaxed);
            const int rc = run_pipe(s, &slots[(size_t)p * (size_t)ring], ring, job, (uint64_t)p, (uint64_t)pipes);
            rcs[(size_t)p] = rc == kStopped ? MODGPU_OK : rc; // (stopped because a sibling failed: that one carries the error)
            if (rcs[(size_t)p]) errs[(size_t)p] = t_err;
            std::lock_guard<std::mutex> lock(mu);
            if (++finished == pipes) cv.notify_all();
        }
    }
    void wait()
    {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return finished == pipes; });
    }
};

// A parked worker: bound to its staging set's NUMA node and to the HIP device once, then serves whatever calls post.
void worker_main(Staging *s, int logical, int phys, cpu_set_t allowed, bool have_allowed)
{
    // a staging worker's copies run where its set's slots are: next to the GPU, or on the node the set was made for
    const int node = s->node.load(std::memory_order_relaxed);
    if (node >= 0 && have_allowed) (void)numa::move_to_node(node, &allowed, sizeof allowed);
    else if (node < 0) run_near_device(logical);
    (void)hipSetDevice(phys); // HIP's current device is per thread
    std::unique_lock<std::mutex> lock(s->mu);
    for (;;) {
        ++s->parked;
        s->work_cv.wait(lock, [&] { return !s->requests.empty(); });
        --s->parked;
        std::shared_ptr<Call> call = std::move(s->requests.front());
        s->requests.pop_front();
        lock.unlock();
        call->help(true);
        call.reset();
        lock.lock();
    }
}

// Makes sure the staging set has `want` worker threads (never more than kMaxPipes - 1 per set) and returns how many it has.  A
// thread that cannot be started (a pids or NPROC limit) is not an error: the caller runs more of the pipelines itself, one after
// another -- which every launch-per-chunk route is fine with, and the host-fed routes are NOT (their kernel draws chunks in stream
// order and waits for whichever pipeline owns the next one: pipelines that run one after another would leave it waiting out its
// patience).  So stream_impl asks BEFORE it decides on a host-fed route.
#ifdef MODGPU_TESTING_HOOKS
std::atomic<int> g_forbid_spawn{0}; // modgpu_debug_forbid_worker_threads
#endif
int ensure_workers(Staging &s, int want, int logical, int phys, const cpu_set_t &mask, bool have_mask)
{
#ifdef MODGPU_TESTING_HOOKS
    if (g_forbid_spawn.load(std::memory_order_relaxed)) return 0; // "no thread can be had": nothing is started, nothing posted
#endif
    std::lock_guard<std::mutex> lock(s.mu);
    want = std::min(want, kMaxPipes - 1);
    while (s.workers < want) {
        try {
            std::thread(worker_main, &s, logical, phys, mask, have_mask).detach();
            ++s.workers;
            g_pool_spawned.fetch_add(1, std::memory_order_relaxed);
        } catch (...) {
            break;
        }
    }
    return s.workers;
}

// Posts `extra` pipelines of `call` to the set's workers (ensure_workers has started them; entries no worker picks up are run by the
// caller itself).
void post_to_workers(Staging &s, const std::shared_ptr<Call> &call, int extra)
{
    if (extra <= 0) return;
    {
        std::lock_guard<std::mutex> lock(s.mu);
        for (int k = 0; k < extra; ++k) s.requests.push_back(call);
    }
    s.work_cv.notify_all();
}

} // namespace

namespace {
// The host loop over the pieces of a lost call whose result has not reached the destination (adjacent ones as one run, so that
// the loop's threads see large spans).  A piece's source bytes are read again where the destination does not still hold them.
int finish_on_host(const Job &j, uint64_t *bytes_done)
{
    const bool in_place = j.src.mem && j.src.mem == j.dst.mem;
    const bool identity = (int64_t)j.key % 0x7FFFFFFFll == 0;
    std::vector<Piece> runs;
    for (size_t k = 0; k < j.plan.size(); ++k) {
        if (j.done[k].load(std::memory_order_acquire)) continue;
        if (!runs.empty() && runs.back().off + runs.back().len == j.plan[k].off) runs.back().len += j.plan[k].len;
        else runs.push_back(j.plan[k]);
    }
    std::vector<uint8_t> tmp;
    for (const Piece &r : runs) {
        if (j.dst.mem) {
            uint8_t *at = j.dst.mem + r.off;
            if (!in_place) { // a file, or other memory: the piece's plaintext comes from there again
                int rc = fill_slot(j.src, at, r.off, r.len);
                if (rc) return rc;
            }
            if (!identity) modgpu_scalar_cycle(at, r.len, j.key, j.stream_off + r.off);
        } else { // the destination is a file: through a bounce buffer, a slot's worth at a time
            tmp.resize((size_t)std::min<uint64_t>(r.len, kChunk));
            for (uint64_t o = 0; o < r.len; o += tmp.size()) {
                const uint64_t l = std::min<uint64_t>(tmp.size(), r.len - o);
                int rc = fill_slot(j.src, tmp.data(), r.off + o, l);
                if (rc) return rc;
                if (!identity) modgpu_scalar_cycle(tmp.data(), l, j.key, j.stream_off + r.off + o);
                rc = drain_slot(j.dst, tmp.data(), r.off + o, l);
                if (rc) return rc;
            }
        }
        *bytes_done += r.len;
    }
    trace(MODGPU_TRACE_RESCUED, -1, runs.size(), *bytes_done);
    return MODGPU_OK;
}
} // namespace

// ---- one call -------------------------------------------------------------------------------------------------------------------
// stream_impl: resolve the device and the staging set, then one of three shapes of call:
//   in_place_on_locked_pages   page-locked caller memory, in place: ONE kernel across PCIe on the pages themselves
//   one_slot_call              header-sized buffers (what the reference's three call sites pass): one slot, one launch
//   pipelined_call             everything else: the stream cut into pieces for up to kPipes pipelines on one of the five routes
//                              above (plan_route says which), then -- if the GPU was lost under way -- the rescue
namespace {
struct CallCtx {
    const Endpoint &src, &dst;
    const uint64_t n;
    const int32_t key;
    const uint64_t stream_off;
    const int dev;
    Staging &s;
    const bool identity, in_place, src_direct, dst_direct, all_direct, host_may_finish;
    int copy_node = -1;
    cpu_set_t caller_mask;
    bool have_mask = false;
    StreamOutcome &outcome;
    bool through_slots = false; // the bytes pass through a page-locked slot with a host copy on a memory side (modgpu_path_stats: staged_bytes)
    void account(uint64_t host_bytes = 0) const
    {
        g_stats.gpu_calls.fetch_add(1, std::memory_order_relaxed);
        g_stats.gpu_bytes.fetch_add(n - host_bytes, std::memory_order_relaxed);
        (all_direct && !through_slots ? g_stats.direct_bytes : g_stats.staged_bytes).fetch_add(n - host_bytes, std::memory_order_relaxed);
    }
};
int injected_here() { return fail(MODGPU_ERR_HIP, "injected failure (modgpu_debug_inject_failure_at)"); }

// Page-locked caller memory of any size, cycled where it lies (default for such memory: 50 GB/s of payload against 26-29 for the
// DMA ring and 30 for the staged route, profiles/r02_sweep_pinned_routes.txt; pinned mode 1 of the testing flavour keeps the DMA ring
// selectable).  One launch + one sync; the kernel reads and writes the pages across PCIe itself (they are device-visible).
int in_place_on_locked_pages(CallCtx &c)
{
    SlotLease lease(c.s, c.dev);
    lease.acquire(1, 1);
    if (lease.ids.empty()) return fail_lost();
    const int slot = lease.ids[0];
    int rc = staging_reserve(c.s, lease.ids, 0, false, false);
    if (rc) return rc;
    void *mapped = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&mapped, c.src.mem, 0));
    if (injected_at(0, 1, MODGPU_STAGE_FILL) || injected_at(0, 1, MODGPU_STAGE_LAUNCH)) return injected_here();
    rc = cycle_device_impl(mapped, c.n, c.key, c.stream_off, c.s.stream[slot], /*over_pcie=*/true);
    if (rc) return rc; // nothing was launched: the caller's pages are as they were
    trace(MODGPU_TRACE_LAUNCHED, -1, 0, c.n);
    // From here on the kernel writes the caller's pages itself.  If the wait for it fails nobody knows which of them it
    // reached before it died, and the plaintext exists nowhere else: THIS route cannot be finished by the host loop, the error
    // stands (include/modgpu.h says so at modgpu_cycle_auto_host).
    c.outcome.touched = true;
    hipError_t e = injected_at(0, 1, MODGPU_STAGE_SYNC) ? hipErrorLaunchFailure : hipStreamSynchronize(c.s.stream[slot]);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(c.s.stream[slot]);
        return fail_hip(e, "hipStreamSynchronize (kernel over PCIe on the caller's page-locked memory)");
    }
    c.account();
    return MODGPU_OK;
}

// Header-sized buffers (<= zero_copy_max): one slot, one kernel across PCIe on the slot, no workers.  The caller's bytes change
// only after the wait has succeeded: the slot takes whatever damage a dying kernel does.
int one_slot_call(CallCtx &c)
{
    Staging &s = c.s;
    SlotLease lease(s, c.dev);
    lease.acquire(1, 1);
    if (lease.ids.empty()) return fail_lost();
    const int slot = lease.ids[0];
    int rc = staging_reserve(s, lease.ids, c.n, false, true);
    if (rc) return rc;
    if (s.pinned_cap[slot] < c.n) return fail(MODGPU_ERR_INVALID, "staging slot smaller than the zero-copy buffer");
    void *mapped = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&mapped, s.pinned[slot], 0));
    if (injected_at(0, 1, MODGPU_STAGE_FILL)) return injected_here();
    std::memcpy(s.pinned[slot], c.src.mem, c.n);
    rc = injected_at(0, 1, MODGPU_STAGE_LAUNCH) ? injected_here() : cycle_device_impl(mapped, c.n, c.key, c.stream_off, s.stream[slot], /*over_pcie=*/true);
    hipError_t e = hipStreamSynchronize(s.stream[slot]);
    if (rc) return rc;
    if (e == hipSuccess && (injected_at(0, 1, MODGPU_STAGE_SYNC) || injected_at(0, 1, MODGPU_STAGE_DRAIN))) e = hipErrorLaunchFailure;
    if (e != hipSuccess) return fail_hip(e, "hipStreamSynchronize (kernel over PCIe)"); // the caller's buffer is as it was
    c.outcome.touched = true;
    std::memcpy(c.dst.mem, s.pinned[slot], c.n);
    c.account();
    return MODGPU_OK;
}

// Which route a pipelined call takes and the size of its pieces.
struct RoutePlan {
    Route route;
    uint64_t chunk;
    bool memory_schedule; // cut and queued like a memory-to-memory call: ~64 chunks of >= 1 MiB, ramped, kernels on shared lanes
};
RoutePlan plan_route(CallCtx &c)
{
    const Endpoint &src = c.src, &dst = c.dst;
    const uint64_t n = c.n;
    // slot size: the whole buffer if it is small, else ~n/split between chunk_min and the cap (8 MiB by default).
    // Memory on both sides, and file -> memory: ~64 chunks of >= 1 MiB, ramped, kernels on shared lanes (profiles/r05_pcie_grid.txt).
    // Memory -> file and file -> file: ~16 chunks of >= 4 MiB, all alike, a stream per slot -- there the slow stage is pwrite,
    // which wants few large calls (profiles/r04_file_routes.txt, r05_file_routes.txt).
    RoutePlan p{};
    p.memory_schedule = (src.mem && dst.mem) || (kFileSched != 0 && !src.mem && dst.mem);
    const uint64_t split = p.memory_schedule ? kSplit : 16, chunk_min = p.memory_schedule ? kChunkMin : std::min<uint64_t>(4ull << 20, kChunk);
    p.chunk = n <= chunk_min ? std::max<uint64_t>(n, 1ull << 20) : std::min<uint64_t>(kChunk, std::max<uint64_t>(chunk_min, ((n / split) + 0xFFFFF) & ~0xFFFFFull));
    p.chunk = std::min<uint64_t>(p.chunk, kChunk);
    // Default routes (profiles/r03_file_routes.txt, r06_file_routes.txt): pageable memory and files are copied / read into a pinned
    // slot and cycled there across PCIe; page-locked caller memory that ends in a file is DMA'd.  (Testing flavour, staged mode 1: the
    // DMA form -- H2D, kernel in HBM, D2H -- of the first.)  A part file that ends in PAGE-LOCKED memory: below 2 GiB through the slots
    // like any other destination (the host-fed kernel wins there); from 2 GiB up -- or with file_feed off -- read straight into that
    // memory and cycled where it lies, a launch per chunk.
    const bool file_to_locked_pages = c.dst_direct && !src.mem && !c.identity && staged_mode() != 1;
    const bool through_slots = file_to_locked_pages && kFeed != 0 && kFileFeed != 0 && staged_mode() == 0 && n < kFeedBelow;
    p.route = file_to_locked_pages && !through_slots                               ? Route::in_dst
              : !c.src_direct && (!c.dst_direct || through_slots) && staged_mode() != 1 ? Route::slot_kernel
                                                                                       : Route::dma;
    // A pageable (or file) source that ends in memory: ONE host-fed kernel for the whole call (cycle_feed_kernel.h) instead of a launch
    // per chunk.  Uniform chunks -- as small as the flag words allow, since no chunk costs a launch -- of whole pieces; a call too large
    // for that (or staged mode 1 / 2 of the testing flavour, or feed switched off) keeps the launch-per-chunk schedule.  The kernel draws
    // chunks in stream order and waits for whichever pipeline owns the next one, so the pipelines must really run side by side: a
    // staging set that cannot have its worker threads does not take this route.
    const bool feedable = kFeed != 0 && staged_mode() == 0 && !c.identity && dst.mem && (src.mem ? !c.src_direct && !c.dst_direct : kFileFeed != 0);
    if (feedable && p.route == Route::slot_kernel) {
        const uint64_t piece = kFeedPieceBytes;
        // (up to 8 MiB half-size chunks: the first copy in and the last copy out are what such a call waits for -- 4 MiB 25.3 -> 27.3,
        //  8 MiB 32.1 -> 33.2 GB/s median, level from 16 MiB; quarter-size chunks lose 10-25 % to the per-chunk flag traffic)
        const uint64_t want = n <= (8ull << 20) ? std::max<uint64_t>(kFeedChunk / 2, piece) : kFeedChunk;
        const uint64_t fc = std::max<uint64_t>((want + piece - 1) / piece * piece, ((n + kFeedChunksMax - 1) / kFeedChunksMax + piece - 1) / piece * piece);
        const int pipes_wanted = (int)std::min<uint64_t>((uint64_t)kPipes, ((n + fc - 1) / fc + 1) / 2);
        if (fc <= kChunk && (n + piece - 1) / piece < kFeedPiecesMax && n < kFeedBelow &&
            (pipes_wanted <= 1 || ensure_workers(c.s, pipes_wanted - 1, c.dev, physical_of(c.dev), c.caller_mask, c.have_mask) >= pipes_wanted - 1)) {
            p.chunk = fc;
            p.route = Route::feed;
        } else if (through_slots) p.route = Route::in_dst; // (no host-fed kernel to be had: the destination itself, as from 2 GiB up)
    }
    return p;
}

// What a host-fed call needs around its pipelines: flag words cleared and handed to the job BEFORE the pipelines start, the one
// launch, and -- whichever way the call ends -- the kernel gone and its counters back at zero before the slots are given back.
struct FeedCall {
    CallCtx &c;
    Job &job;
    const std::vector<int> &slots;
    const RoutePlan &plan;
    int pipes = 0, lead_slot = -1;
    uint64_t chunks = 0;
    int prepare(int pipes_)
    {
        Staging &s = c.s;
        pipes = pipes_;
        lead_slot = slots[0];
        const int rc = feed_reserve(s, lead_slot);
        if (rc) return rc;
        chunks = job.plan.size();
        uint32_t *const flags = s.feed_flags[lead_slot];
        std::memset(flags, 0, chunks * sizeof(uint32_t));
        std::memset(flags + kFeedChunksMax, 0, chunks * sizeof(uint32_t));
        flags[2 * kFeedChunksMax] = 0;
        job.feed_ready = flags;
        job.feed_done = flags + kFeedChunksMax;
        job.feed_abort = flags + 2 * kFeedChunksMax;
        job.feed_stream = s.stream[lead_slot];
        return MODGPU_OK;
    }
    // (~15 us of host time: made after the workers have been posted, while every pipeline copies its first chunk in -- the kernel is
    //  there by the time the first chunk is marked ready, and a 4 MiB call is 20 us shorter than with the launch in front,
    //  profiles/r05_pcie_feed.txt)
    int launch()
    {
        Staging &s = c.s;
        CycleFeedArgs a{};
        for (int k = 0; k < pipes * 2; ++k) HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&a.slot[k]), s.pinned[slots[(size_t)k]], 0));
        a.ready = s.feed_flags_dev[lead_slot];
        a.done = s.feed_flags_dev[lead_slot] + kFeedChunksMax;
        a.abort = s.feed_flags_dev[lead_slot] + 2 * kFeedChunksMax;
        a.work = s.feed_work[lead_slot]; // (all zero: feed_reserve, finish)
        a.n = c.n;
        a.patience_ticks = kFeedPatienceTicks;
        a.chunk_bytes = (uint32_t)plan.chunk;
        a.pipes = (uint32_t)pipes;
        a.base = lcg::state_residue(lcg::key_residue(c.key), c.stream_off);
        const uint32_t pieces = (uint32_t)((a.n + kFeedPieceBytes - 1) / kFeedPieceBytes);
        const uint32_t grid = std::min<uint32_t>(kFeedGrid, pieces);
        const hipError_t e = modgpu_launch_cycle_feed(a, grid, job.feed_stream);
        if (e != hipSuccess) return fail_hip(e, "cycle kernel launch (host-fed)");
        job.feed_launched.store(true, std::memory_order_release);
        note_feed_launch(grid, c.n);
        trace(MODGPU_TRACE_LAUNCHED, -1, 0, c.n);
        return MODGPU_OK;
    }
    void finish(int rc)
    {
        Staging &s = c.s;
        // rc == OK: every chunk is done and drained, the kernel has drawn its last ticket and leaves by itself (an error of the stream here
        // cannot undo the result, which is whole in the destination: the next call on this device meets whatever is wrong with it).
        // Otherwise whichever pipeline failed has told the kernel to leave and waited for it; make sure before the slots go back.  Either way
        // the wait is bounded: a kernel that does not leave costs the device its host-buffer routes (feed_stop), not this thread.
        const std::string keep = t_err;
        const bool gone = feed_stop(job);
        (void)hipGetLastError();
        t_err = keep;
        if (!gone) return; // (slots, flag words and counters stay with the kernel that would not leave: SlotLease keeps them out of circulation)
        // the counters go back to zero behind the call (asynchronously, on the slot's own stream: in front of its next launch)
        if (hipMemsetAsync(s.feed_work[lead_slot], 0, (chunks + 2) * sizeof(uint32_t), job.feed_stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(s.feed_work[lead_slot]); // (cannot be trusted any more: the next call that leads with this slot makes new ones)
            s.feed_work[lead_slot] = nullptr;
        }
    }
};

// The GPU was lost after the call had begun.  Every pipeline has stopped and waited for what it had in flight.  A piece is either
// marked done -- its result is in the destination, whole -- or not, and then the host loop can still do it as long as its plaintext
// is still somewhere: in the source file, in the caller's other buffer, or in the destination itself when that is only ever written
// by a finished piece's copy out of its slot (the staged routes).  In place AND written directly by the device (page-locked memory
// in DMA mode) the plaintext of an unfinished piece is gone: the error stands.
int rescue_on_host(CallCtx &c, const Job &job, int rc)
{
    const bool recoverable = rc == MODGPU_ERR_HIP && !(c.in_place && job.dst_written_by_device());
    if (!c.host_may_finish || !recoverable) return rc;
    const std::string why = t_err;
    uint64_t host_bytes = 0;
    const int rc2 = finish_on_host(job, &host_bytes);
    if (rc2 != MODGPU_OK) return rc2; // (an I/O error of the rescue itself: its own text)
    c.outcome.finished_on_host = true;
    c.outcome.host_bytes = host_bytes;
    c.outcome.touched = true;
    c.account(host_bytes);
    g_stats.midcall_rescues.fetch_add(1, std::memory_order_relaxed);
    g_stats.midcall_rescued_bytes.fetch_add(host_bytes, std::memory_order_relaxed);
    g_stats.scalar_bytes.fetch_add(host_bytes, std::memory_order_relaxed);
    t_err = "finished on the host loop after: " + why;
    return MODGPU_OK;
}

int pipelined_call(CallCtx &c)
{
    Staging &s = c.s;
    const RoutePlan plan = plan_route(c);
    const bool feed = plan.route == Route::feed;
    const uint64_t n_chunks = (c.n + plan.chunk - 1) / plan.chunk;
    Job job(c.src, c.dst, c.n, plan.chunk, c.key, c.stream_off);
    job.route = plan.route;
    job.dev = c.dev;
    c.through_slots = plan.route == Route::feed || plan.route == Route::slot_kernel;
    int pipes, ring;
    if (c.all_direct && c.src.mem && c.dst.mem) { // no host work at all: one thread keeps a ring of slots busy
        pipes = 1;
        ring = (int)std::min<uint64_t>((uint64_t)kRing, std::max<uint64_t>(n_chunks, 2));
    } else {
        pipes = (int)std::min<uint64_t>((uint64_t)kPipes, (n_chunks + 1) / 2); // a pipeline is worth >= 2 chunks
        ring = 2;
    }
    // a destination file gets its blocks before eight threads write into it at once (tmpfs and most file systems
    // allocate under one lock: parallel extending writes serialise there)
    if (c.dst.fd >= 0 && c.n >= (8ull << 20)) (void)::posix_fallocate(c.dst.fd, (off_t)c.dst.base, (off_t)c.n);
    // this call's slots: what it would like, or as many whole pipelines as are free right now (another caller may be at work
    // on this GPU), at least one
    SlotLease lease(s, c.dev);
    lease.acquire(pipes * ring, ring);
    if (lease.ids.empty()) return fail_lost();
    pipes = (int)lease.ids.size() / ring;
    job.set_plan(cut_stream(c.n, plan.chunk, pipes, plan.memory_schedule && !feed ? kRamp : 0));
    job.copy_node = c.copy_node;
    job.have_mask = c.have_mask;
    if (c.have_mask) job.caller_mask = c.caller_mask;
    for (int k = 0; k < (feed ? 0 : plan.memory_schedule ? kLanes : kFileLanes) && k < (int)lease.ids.size(); ++k) job.lanes.push_back(nullptr); // (streams exist after staging_reserve)
    trace(MODGPU_TRACE_SLOTS, -1, (uint64_t)pipes, plan.chunk);
    int rc = staging_reserve(s, lease.ids, plan.chunk, plan.route == Route::dma, !(c.src_direct && c.dst_direct) && plan.route != Route::in_dst);
    if (rc) return rc;
    for (size_t k = 0; k < job.lanes.size(); ++k) job.lanes[k] = s.stream[lease.ids[k]];
    FeedCall fed{c, job, lease.ids, plan};
    if (feed) {
        rc = fed.prepare(pipes);
        if (rc) return rc;
    }

    if (pipes <= 1) {
        if (feed) {
            rc = fed.launch();
            if (rc) return rc; // nothing of the caller's has been touched
        }
        rc = run_pipe(s, lease.ids.data(), ring, job, 0, 1);
        if (rc == kStopped) rc = MODGPU_OK;
    } else {
        auto call = std::make_shared<Call>(s, job, pipes, ring, physical_of(c.dev), lease.ids);
        const int workers = ensure_workers(s, pipes - 1, c.dev, physical_of(c.dev), c.caller_mask, c.have_mask);
        post_to_workers(s, call, std::min(pipes - 1, workers)); // (pipelines nobody is there for are run by this thread, after its own)
        trace(MODGPU_TRACE_POSTED, -1, (uint64_t)pipes, 0);
        int rc_launch = MODGPU_OK;
        std::string launch_err;
        if (feed) {
            rc_launch = fed.launch();
            if (rc_launch) { // no kernel: the pipelines stop at their next step; none has drained anything (nothing was ever marked done)
                launch_err = t_err;
                job.failed.store(true, std::memory_order_release);
            }
        }
        call->help(false); // pipeline 0 starts now, on the calling thread; then whatever no worker has picked up yet
        call->wait();
        { // entries of this call that no worker has picked up are of no use to anybody now
            std::lock_guard<std::mutex> lock(s.mu);
            s.requests.erase(std::remove(s.requests.begin(), s.requests.end(), call), s.requests.end());
        }
        for (int p = 0; p < pipes && rc == MODGPU_OK; ++p)
            if (call->rcs[(size_t)p]) {
                t_err = call->errs[(size_t)p];
                rc = call->rcs[(size_t)p];
            }
        if (rc_launch) {
            t_err = launch_err;
            rc = rc_launch;
        }
    }
    c.outcome.touched = job.touched.load();
    if (feed) fed.finish(rc);
    if (rc == MODGPU_OK) {
        c.account();
        return rc;
    }
    return rescue_on_host(c, job, rc);
}
} // namespace

int stream_impl(const Endpoint &src, const Endpoint &dst, uint64_t n, int32_t key, uint64_t stream_off, int device,
                bool host_may_finish, StreamOutcome *out)
{
    StreamOutcome dummy;
    StreamOutcome &outcome = out ? *out : dummy;
    outcome = StreamOutcome{};
    if (n == 0) return MODGPU_OK;
    // byte j is at stream position stream_off + j, taken in the integers and reduced mod the generator's period: reduce the
    // offset first, so that adding a chunk's position below can never wrap at 2^64 (which is not a multiple of the period)
    stream_off %= 0x7FFFFFFEull;
    int dev = 0;
    DeviceScope scope(device); // the calling thread gets its own current device back
    int rc = scope.rc ? scope.rc : resolve_device(device, &dev);
    if (rc) return rc;
    if (injected_failure()) return fail(MODGPU_ERR_HIP, "injected failure (modgpu_debug_inject_failures)");
    // keys == 0 mod m give the identity: nothing to do in place, a plain copy otherwise
    const bool identity = (int64_t)key % 0x7FFFFFFFll == 0;
    const bool in_place = src.mem && src.mem == dst.mem;
    if (identity && in_place) return MODGPU_OK;
    if (dev >= kMaxDevices) return fail(MODGPU_ERR_INVALID, "device index beyond staging table");
    if (g_device_lost[dev].load(std::memory_order_acquire)) return fail_lost(); // (before anything of the caller's is touched)
    const bool src_direct = src.mem && src.pinned, dst_direct = dst.mem && dst.pinned;
    // which of the device's staging sets: the one on the node the caller's PAGEABLE pages live on, if that is not the GPU's own
    int copy_node = -1;
    cpu_set_t caller_mask;
    CPU_ZERO(&caller_mask);
    bool have_mask = false;
    if (numa::enabled() && n > kZeroCopyMax) {
        // (what decides is the side the CPU READS across the socket link: a copy whose source is remote runs at 6-12 GB/s per thread
        //  instead of 28, one whose destination is remote hardly suffers, profiles/r05_staged_numa.txt)
        const uint8_t *pages = src.mem && !src_direct ? src.mem : dst.mem && !dst_direct ? dst.mem : nullptr;
        if (!src.mem && dst.mem) { // a part file read into memory: where the file's cached pages are (pread is the remote read then)
            copy_node = numa::node_of_file_page(src.fd, src.base + n / 2);
            if (copy_node < 0 && pages) copy_node = numa::node_of_address(pages + n / 2);
        } else if (pages) copy_node = numa::node_of_address(pages + n / 2);
        if (copy_node >= 0) have_mask = ::sched_getaffinity(0, sizeof caller_mask, &caller_mask) == 0;
    }
    const int gpu_node = copy_node >= 0 ? device_numa_node(dev) : -1;
    const int set = copy_node >= 0 && copy_node < kNodeSets - 1 && gpu_node >= 0 && copy_node != gpu_node && have_mask ? 1 + copy_node : 0;
    Staging &s = staging_of(dev, set);
    if (set != 0) { // (every caller of this set writes the same value: a set belongs to one node)
        if (s.node.load(std::memory_order_relaxed) != copy_node) s.node.store(copy_node, std::memory_order_relaxed);
        g_node_set_calls.fetch_add(1, std::memory_order_relaxed);
    }
    struct InFlight { // (reporting: did host-buffer calls ever overlap on a GPU?  modgpu_host_pool_stats)
        InFlight() { if (g_calls_in_flight.fetch_add(1, std::memory_order_relaxed) > 0) g_calls_overlapped.fetch_add(1, std::memory_order_relaxed); }
        ~InFlight() { g_calls_in_flight.fetch_sub(1, std::memory_order_relaxed); }
    } in_flight;
    trace(MODGPU_TRACE_CALL_BEGIN, -1, 0, n);
    struct CallEnd { uint64_t n; ~CallEnd() { trace(MODGPU_TRACE_CALL_END, -1, 0, n); } } call_end{n};

    CallCtx c{src, dst, n, key, stream_off, dev, s, identity, in_place, src_direct, dst_direct,
              (!src.mem || src_direct) && (!dst.mem || dst_direct), host_may_finish, copy_node, caller_mask, have_mask, outcome};
    if (in_place && src_direct && !identity && (n <= kZeroCopyMax || pinned_mode() != 1)) return in_place_on_locked_pages(c);
    if (n <= kZeroCopyMax && src.mem && dst.mem && !identity) return one_slot_call(c);
    return pipelined_call(c);
}

} // namespace modgpu

// ---- file endpoints of the ABI -----------------------------------------------------------------
using namespace modgpu;

namespace {
struct Fd { // closes on scope exit
    int fd = -1;
    ~Fd() { if (fd >= 0) ::close(fd); }
};
} // namespace

extern "C" {

#ifdef MODGPU_TESTING_HOOKS
void modgpu_debug_inject_failures(int count) { g_inject_failures.store(count > 0 ? count : 0); }
void modgpu_debug_inject_failure_at(int64_t piece, int stage)
{
    g_inject_stage.store(-1, std::memory_order_release);
    g_inject_piece.store(piece, std::memory_order_relaxed);
    g_inject_stage.store(stage >= MODGPU_STAGE_FILL && stage <= MODGPU_STAGE_STALL ? stage : -1, std::memory_order_release);
}
void modgpu_debug_forbid_worker_threads(int forbid) { g_forbid_spawn.store(forbid ? 1 : 0); }
int modgpu_debug_injection_armed(void) { return g_inject_stage.load(std::memory_order_acquire) >= 0 ? 1 : 0; }
// Takes `count` PIPELINE slots of a device's own staging set, as large calls do, and keeps them until called with count = 0: with all
// of them held, what a header-sized call still finds is exactly the slots reserved for it.  Returns how many are held now.
int modgpu_debug_hold_slots(int device, int count)
{
    static std::mutex mu;
    static std::unique_ptr<SlotLease> held[kMaxDevices];
    if (device < 0 || device >= kMaxDevices) return -1;
    std::lock_guard<std::mutex> lock(mu);
    held[device].reset();
    if (count <= 0) return 0;
    held[device].reset(new SlotLease(staging_of(device, 0), device));
    held[device]->acquire(count, 2);
    return (int)held[device]->ids.size();
}
// Run-time form of the testing flavour's knobs; not while a host-buffer call is in flight.  Values are clamped like the environment's.
void modgpu_debug_set_host_tunable(int which, uint64_t value)
{
    auto clamp = [](uint64_t v, uint64_t lo, uint64_t hi) { return v < lo ? lo : (v > hi ? hi : v); };
    switch (which) {
    case MODGPU_TUNABLE_ZEROCOPY_BYTES: kZeroCopyMax = std::min<uint64_t>(value, kChunk); break;
    case MODGPU_TUNABLE_RING: kRing = (int)clamp(value, 2, 4); break;
    case MODGPU_TUNABLE_SPLIT: kSplit = clamp(value, 2, 256); break;
    case MODGPU_TUNABLE_CHUNK_MIN_BYTES: kChunkMin = std::min<uint64_t>(clamp(value, 1ull << 20, 256ull << 20), kChunk); break;
    case MODGPU_TUNABLE_RAMP_BYTES: kRamp = std::min<uint64_t>(value, kChunk); break;
    case MODGPU_TUNABLE_LANES: kLanes = (int)clamp(value, 0, 8); break;
    case MODGPU_TUNABLE_NTCOPY: kNtCopy = value != 0; break;
    case MODGPU_TUNABLE_FILE_SCHED: kFileSched = value != 0; break;
    case MODGPU_TUNABLE_FEED: kFeed = value != 0; break;
    case MODGPU_TUNABLE_FEED_CHUNK_BYTES: kFeedChunk = clamp(value, 32ull << 10, 8ull << 20); break;
    case MODGPU_TUNABLE_FILE_FEED: kFileFeed = value != 0; break;
    case MODGPU_TUNABLE_FEED_PATIENCE_MS: kFeedPatienceTicks = clamp(value, 1, 600000) * 100000ull; break;
    default: break;
    }
}
#endif

void modgpu_host_trace(int enable)
{
    if (enable) {
        std::lock_guard<std::mutex> lock(g_trace_mu);
        g_trace.clear();
        g_trace.reserve(1u << 16);
    }
    g_trace_on.store(enable != 0, std::memory_order_relaxed);
}

int modgpu_host_trace_read(modgpu_host_trace_event_t *out, int cap)
{
    std::lock_guard<std::mutex> lock(g_trace_mu);
    const int n = (int)std::min<size_t>(g_trace.size(), (size_t)(cap > 0 ? cap : 0));
    if (out && n > 0) std::memcpy(out, g_trace.data(), (size_t)n * sizeof(modgpu_host_trace_event_t));
    return (int)g_trace.size();
}

void modgpu_host_pool_stats(uint64_t out[6])
{
    out[0] = g_pool_spawned.load();
    out[1] = g_pool_tasks.load();
    out[2] = g_slot_waits.load();
    out[3] = g_calls_overlapped.load();
    out[4] = (uint64_t)kSlots;
    out[5] = g_node_set_calls.load();
}

void modgpu_host_chunking(uint64_t out[4])
{
    out[0] = kSplit;
    out[1] = kChunkMin;
    out[2] = kRamp;
    out[3] = (uint64_t)kLanes;
}

void modgpu_host_tunables(uint64_t out[4])
{
    out[0] = (uint64_t)kPipes;
    out[1] = kChunk;
    out[2] = kZeroCopyMax;
    out[3] = (uint64_t)kRing;
}

int modgpu_cycle_file(const char *src_path, const char *dst_path, int32_t key, uint64_t stream_off, int device)
{
    return guarded([&]() -> int {
        if (!src_path || !dst_path) return fail(MODGPU_ERR_INVALID, "null path");
        // dst is opened WITHOUT truncation first: if it turns out to be the source under another
        // spelling (./a vs a, a symlink, a hard link) truncating it would destroy the input.
        Fd in, out;
        in.fd = ::open(src_path, O_RDONLY);
        if (in.fd < 0) return fail_io(src_path);
        struct stat st_in, st_out;
        if (::fstat(in.fd, &st_in) != 0) return fail_io("fstat");
        out.fd = ::open(dst_path, O_RDWR | O_CREAT, 0644);
        if (out.fd < 0) return fail_io(dst_path);
        if (::fstat(out.fd, &st_out) != 0) return fail_io("fstat");
        const bool in_place = st_in.st_dev == st_out.st_dev && st_in.st_ino == st_out.st_ino;
        if (!in_place && ::ftruncate(out.fd, 0) != 0) return fail_io("ftruncate");
        Endpoint src, dst;
        src.fd = in_place ? out.fd : in.fd;
        dst.fd = out.fd;
        // (a GPU lost after the call has begun: the source file still holds every byte -- in place, too, since a piece is written
        //  only when it is finished -- so the host loop does the pieces that have not been written; see modgpu_cycle_file_to_host)
        return stream_impl(src, dst, (uint64_t)st_in.st_size, key, stream_off, device, !gpu_required(), nullptr);
    });
}

int modgpu_cycle_file_to_host(const char *path, uint64_t file_off, uint8_t *host_dst, uint64_t n, int32_t key,
                              uint64_t stream_off, int device)
{
    return guarded([&]() -> int {
        if (!path || (n && !host_dst)) return fail(MODGPU_ERR_INVALID, "null path or buffer");
        Fd in;
        in.fd = ::open(path, O_RDONLY);
        if (in.fd < 0) return fail_io(path);
        Endpoint src, dst;
        src.fd = in.fd;
        src.base = file_off;
        dst.mem = host_dst;
        dst.pinned = host_range_pinned(host_dst, n);
        // The file still holds every byte: if the GPU is lost after the call has begun, the pieces that have not arrived are
        // read again and done by the host loop (unless MODGPU_REQUIRE_GPU=1) -- LoadArkData's part cipher keeps going.
        return stream_impl(src, dst, n, key, stream_off, device, !gpu_required(), nullptr);
    });
}

int modgpu_cycle_host_to_file(const uint8_t *host_src, uint64_t n, const char *path, int32_t key, uint64_t stream_off,
                              int device)
{
    return guarded([&]() -> int {
        if (!path || (n && !host_src)) return fail(MODGPU_ERR_INVALID, "null path or buffer");
        Fd out;
        out.fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (out.fd < 0) return fail_io(path);
        Endpoint src, dst;
        src.mem = const_cast<uint8_t *>(host_src); // only read from
        src.pinned = host_range_pinned(host_src, n);
        dst.fd = out.fd;
        return stream_impl(src, dst, n, key, stream_off, device, !gpu_required(), nullptr); // (host_src is never modified: lost pieces are done from it again)
    });
}

} // extern "C"
