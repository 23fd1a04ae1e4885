/*
 * modgpu_testing.h -- measurement and test hooks.
 *
 * NOT part of the drop-in boundary (that is include/modgpu.h, the only header an integrator
 * needs).  Two groups:
 *
 *   reporting   in libmodgpu.so.  They time launches on the launch stream and say which kernel
 *               instantiation a launch used, what the library latched from the environment, and what
 *               the work-queue bookkeeping did.  They change nothing.
 *   modgpu_debug_*   ONLY in the testing flavour, libmodgpu_testing.so (the same sources built with
 *               -DMODGPU_TESTING_HOOKS).  They force launch shapes, routes, the size of the ticket ring
 *               and failures, process-wide, so that tests can drive every branch on small buffers.  The
 *               shipped library does not contain them (`nm -D libmodgpu.so | grep debug_` is empty):
 *               nothing in a production process can change how it launches or make a call fail.
 */
#ifndef MODGPU_TESTING_H
#define MODGPU_TESTING_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Runs `iters` back-to-back modgpu_cycle_device launches on `hip_stream` bracketed by HIP events
 * recorded on that same stream and returns the mean milliseconds per launch in *ms_per_launch
 * (an even `iters` leaves the buffer unchanged: the cipher is an involution). */
int modgpu_time_cycle_device(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off,
                             int device, void *hip_stream, int iters, float *ms_per_launch);

/* The launch the calling thread made last (any entry point), as the library planned it. */
typedef struct modgpu_launch_info {
    const char *kernel;   /* the instantiation's name as rocprofv3 prints it, e.g.
                             "modgpu_cycle_queue_kernel<4, 1024>"; static storage */
    int variant;          /* 0 = small shape, 1 = streaming shape (static chunk map), 2 = streaming shape fed by the work queue,
                             3 = the work-queue shape over several parts in one launch (modgpu_cycle_batch_device; `bytes` = all of them),
                             4 = the host-fed kernel of a host-buffer call (one launch for the whole call; `bytes` = the call's) */
    uint32_t grid;        /* workgroups launched                                                  */
    uint32_t block;       /* threads per workgroup                                                */
    uint32_t chunk_bytes; /* bytes one workgroup trip covers                                      */
    uint64_t bytes;       /* n of that launch                                                     */
    uint32_t main_groups; /* of `grid`: workgroups that stream from the start; the other grid - main_groups are helper
                             workgroups of the work-queue shape, which join only while the shader clock is low */
    const char *source_hash; /* identity of the TU that kernel was compiled from: modgpu_kernel_source_hash() for variants 0..3,
                                modgpu_feed_kernel_source_hash() for variant 4; static storage */
} modgpu_launch_info_t;
int modgpu_last_launch(modgpu_launch_info_t *out);

/* The host-path tunables as the library latched them at load (after clamping): out[0] = pipelines
 * (MODGPU_HOST_PIPES), out[1] = largest slot in bytes (MODGPU_HOST_CHUNK_MB), out[2] = largest buffer cycled in
 * one pinned slot without chunking (MODGPU_HOST_ZEROCOPY_KB, never above out[1]), out[3] = DMA ring depth. */
void modgpu_host_tunables(uint64_t out[4]);
/* ... and the three that shape the chunks of a staged buffer: out[0] = MODGPU_HOST_SPLIT (a buffer is cut into about this many
 * chunks), out[1] = smallest such chunk in bytes (MODGPU_HOST_CHUNK_MIN_MB, never above the largest slot), out[2] = size of each
 * pipeline's first and last chunk in bytes (MODGPU_HOST_RAMP_KB; 0 = no ramp), out[3] = streams a call's kernels across PCIe are
 * queued on in launch order (MODGPU_HOST_LANES; 0 = one stream per slot).  In the shipped library these four, the ring depth and
 * the one-slot limit are constants; the environment names work in the testing flavour only. */
void modgpu_host_chunking(uint64_t out[4]);

/* Host-side timeline of the host-buffer / file routes: while enabled, every call of modgpu_cycle_host and
 * the file entry points records what it did and when (CLOCK_MONOTONIC ns): the call's begin and end, the slots it got
 * (chunk = pipelines, bytes = slot size), its pipelines posted to the device's parked workers, and per pipeline and chunk:
 * fill begin / end (memcpy or pread into the slot), the kernel launch returning, the wait for the kernel (sync begin / end),
 * the drain's end (memcpy or pwrite out of the slot).  pipe = -1 for call-level events.  Recording costs a clock read and a
 * short lock per event; disabled (the default) it is one relaxed load.  bin/modbench --hostcall --trace prints it. */
typedef struct modgpu_host_trace_event {
    uint64_t t_ns;
    int kind; /* MODGPU_TRACE_* */
    int pipe;
    uint64_t chunk, bytes;
    int tid;      /* OS thread id of the recording thread: rocprofv3's kernel trace names the launching thread of every dispatch, so
                     a LAUNCHED event and its kernel find each other (tools/summarize_pcie_trace.py) */
    int reserved;
} modgpu_host_trace_event_t;
enum {
    MODGPU_TRACE_CALL_BEGIN = 0, MODGPU_TRACE_SLOTS = 1, MODGPU_TRACE_POSTED = 2, MODGPU_TRACE_PIPE_START = 3, MODGPU_TRACE_FILL_BEGIN = 4,
    MODGPU_TRACE_FILL_END = 5, MODGPU_TRACE_LAUNCHED = 6, MODGPU_TRACE_SYNC_BEGIN = 7, MODGPU_TRACE_SYNC_END = 8, MODGPU_TRACE_DRAIN_END = 9,
    MODGPU_TRACE_PIPE_END = 10, MODGPU_TRACE_CALL_END = 11,
    MODGPU_TRACE_FAILED = 12, /* a pipeline met a failure: chunk = the piece, bytes = the stage (MODGPU_STAGE_*) */
    MODGPU_TRACE_RESCUED = 13, /* the host loop finished the call: chunk = runs of adjacent pieces, bytes = their bytes */
    MODGPU_TRACE_READY = 14 /* host-fed call: a pipeline marked its chunk ready for the call's one kernel (what LAUNCHED is per chunk on the other routes;
                             * the call's single launch is one LAUNCHED event of pipe -1 with the call's bytes) */
};
/* enable != 0 clears the buffer and starts recording (at most 2^20 events are kept); 0 stops. */
void modgpu_host_trace(int enable);
/* Copies the first min(cap, recorded) events to out (may be NULL with cap 0) and returns how many were recorded. */
int modgpu_host_trace_read(modgpu_host_trace_event_t *out, int cap);
/* The staging contexts' bookkeeping since load: out[0] = worker threads started (they park between calls and are never
 * joined), out[1] = pipelines run by workers, out[2] = calls that had to wait for a slot, out[3] = host-buffer calls that began
 * while another was in flight (any device), out[4] = slots per staging set, out[5] = calls that took the staging set of ANOTHER NUMA
 * node than the GPU's because the caller's pageable pages live there (slots allocated on that node, workers bound to it). */
void modgpu_host_pool_stats(uint64_t out[6]);

/* What shapes the host loop's threading in this process: out[0] = most threads per call (MODGPU_HOST_THREADS as latched),
 * out[1] = the control group's CPU limit (cpu.max; 0 = none known), out[2] = CPUs in the calling thread's affinity mask,
 * out[3] = parked worker threads started so far. */
void modgpu_host_loop_info(uint64_t out[4]);

/* modgpu_cycle_scalar_host with one named body ("generic", "avx2", "avx512"); MODGPU_ERR_INVALID if this CPU
 * does not run it.  Lets the tests compare every body with the oracle on one machine. */
int modgpu_cycle_scalar_host_isa(uint8_t *host_buf, uint64_t n, int32_t key, uint64_t stream_off, const char *isa);

/* The topology reader behind modgpu_host_alloc_near, pointed at any sysfs tree (tests hand it a fake one):
 * *node = NUMA node of PCI function `bdf` (-1 unknown), cpus[0..return) = that node's CPUs (at most max_cpus). */
int modgpu_numa_probe(const char *sysfs_root, const char *bdf, int *node, int *cpus, int max_cpus);

/* modgpu_host_alloc_near with the node named outright (measurement: the same kernel on memory next to the GPU and on the other
 * socket's, bin/modbench --numa).  MODGPU_ERR_INVALID if mbind refuses the node.  Free with modgpu_host_free. */
int modgpu_host_alloc_on_node(void **host_ptr, uint64_t n, int node);

/* Work-queue bookkeeping since load: out[0] = eager launches that got a ring line, out[1] = eager launches that
 * found every line busy and took the static shape, out[2] = captured launches that got a line of their own,
 * out[3] = captured launches that found the pool empty (static shape), out[4] = launches that carried several parts
 * (modgpu_cycle_batch_device, modgpu_cycle_parts_device), out[5] = the parts they carried. */
void modgpu_queue_stats(uint64_t out[6]);

/* Identity of the device code this library carries: hex SHA-256 over the kernel sources it was
 * built from (cycle_kernel_impl.h, cycle_kernel.hip, cycle_kernel.h, lcg.h), fixed at build time.
 * profiles/pmc_summary.json records it so that counter figures are never replayed for other code. */
const char *modgpu_kernel_source_hash(void);
/* The same for the host-fed kernel's TU (cycle_feed_kernel.hip, cycle_feed_kernel.h, cycle_kernel_impl.h, lcg.h): every
 * `roofline_pcie` profile of a host-buffer route records it. */
const char *modgpu_feed_kernel_source_hash(void);

/* 1 in libmodgpu_testing.so, 0 in libmodgpu.so. */
int modgpu_testing_hooks(void);

/* ---- below: libmodgpu_testing.so only ------------------------------------------------------------------ */

/* Eager work-queue launches draw their ticket pair from the first `lines` lines of the ring (1..4096; 0 = all
 * 4096).  With one line every second launch in flight finds the ring busy: the collision the gating exists for
 * becomes certain instead of a 1-in-4096 event. */
void modgpu_debug_set_queue_ring(uint32_t lines);

/* Helper workgroups of the work-queue shape (one per CU the main workgroups leave idle; they join only while the shader clock is
 * low): 0 = decide by the clock they measure (the shipped behaviour), 1 = always join, 2 = launch none.  Lets the parity tests
 * run both branches whatever the chip's clock happens to be. */
void modgpu_debug_set_helpers(int mode);

/* modgpu_cycle_batch_device / modgpu_cycle_parts_device: 0 = several parts share a launch when together they are beyond 256 MiB
 * or small on average (the shipped behaviour), 1 = every run of two or more non-empty parts does, 2 = one launch per part. */
void modgpu_debug_set_batch(int mode);

/* Forces the launch shape of every later launch in this process (-1 = by size, the default) and
 * caps the grid (0 = no cap).  Lets the parity tests run the streaming kernels with 1, 2, odd and
 * even trip counts and ragged ends on buffers of a few MiB. */
void modgpu_debug_set_launch(int variant, uint32_t grid_cap);

/* The staging knobs that are constants in the shipped library (host_stream.cpp says which and why): this flavour reads them from
 * the environment at load under their old names (MODGPU_HOST_ZEROCOPY_KB, _RING, _SPLIT, _CHUNK_MIN_MB, _RAMP_KB, _LANES, _NTCOPY)
 * and changes them here at run time -- not while a host-buffer call is in flight.  modgpu_host_tunables / _chunking report them. */
enum { MODGPU_TUNABLE_ZEROCOPY_BYTES = 0, MODGPU_TUNABLE_RING = 1, MODGPU_TUNABLE_SPLIT = 2, MODGPU_TUNABLE_CHUNK_MIN_BYTES = 3,
       MODGPU_TUNABLE_RAMP_BYTES = 4, MODGPU_TUNABLE_LANES = 5, MODGPU_TUNABLE_NTCOPY = 6,
       MODGPU_TUNABLE_FILE_SCHED = 7, /* 1 (the shipped rule): file -> memory is cut and queued like a memory-to-memory call; 0: like the other file routes */
       MODGPU_TUNABLE_FEED = 8,       /* 1 (the shipped rule): pageable memory on both sides is cycled by ONE host-fed kernel per call; 0: a launch per chunk */
       MODGPU_TUNABLE_FEED_CHUNK_BYTES = 9, /* chunk of a host-fed call (256 KiB; whole 32 KiB pieces) */
       MODGPU_TUNABLE_FEED_PATIENCE_MS = 10, /* how long the host-fed kernel waits for one chunk before it gives the call up (10 000) */
       MODGPU_TUNABLE_FILE_FEED = 11 /* 1 (the shipped rule): a FILE that ends in memory (pageable, or page-locked below 2 GiB) takes the host-fed
                                        kernel too, pread in place of the copy into the slot; 0: round 5's launch per chunk */ };
void modgpu_debug_set_host_tunable(int which, uint64_t value);

/* The NUMA node the library believes its GPUs hang off (-1 = unknown, -2 = ask sysfs, the default).  Lets a one-node machine
 * exercise the staging set of "another node than the GPU's" (own placed slots, workers bound to the node). */
void modgpu_debug_set_gpu_node(int node);

/* Workgroups of a launch that works across PCIe on page-locked host memory (0 = the product's rule).  Measurement only
 * (tools/sweep_pcie_grid.py). */
void modgpu_debug_set_pcie_grid(uint32_t cap);

/* How modgpu_cycle_host treats a pinned caller buffer: 0 = library default (= 2), 1 = DMA ring
 * (H2D -> kernel in HBM -> D2H straight from / to the caller's pages), 2 = one kernel over PCIe on
 * the pages themselves.  Both give the same bytes; tools/archive/sweep_pinned.py times them. */
void modgpu_debug_set_pinned_mode(int mode);

/* How a staged chunk (pageable memory or a file, copied into a pinned slot) is cycled: 0 = library
 * default (the kernel works on the pinned slot across PCIe -- no DMA submissions, no device slot --, and where both sides
 * are pageable memory it is ONE host-fed kernel per call, cycle_feed_kernel.h), 1 = H2D -> kernel in HBM -> D2H,
 * 2 = the kernel on the slot with a launch per chunk everywhere (the default until round 5).  Same bytes. */
void modgpu_debug_set_staged_mode(int mode);

/* Failure injection: the next `count` host-buffer / file calls fail with MODGPU_ERR_HIP before they touch
 * anything, as if a HIP call had failed at set-up.  Lets the tests drive modgpu_cycle_auto_host's second branch
 * ("a GPU is visible but the attempt failed") on a machine whose GPU works. */
void modgpu_debug_inject_failures(int count);

/* Failure injection in the MIDDLE of a call: arms one failure -- "the HIP call of `stage` for piece `piece` of the
 * next host-buffer / file call that has such a piece fails with MODGPU_ERR_HIP" -- which fires once.  Pieces are numbered in
 * stream order (a call's plan: host_stream.cpp cut_stream; a header-sized or page-locked in-place call is one piece, number 0);
 * MODGPU_INJECT_PIECE_LAST / _MIDDLE name the last piece and the one at half the plan, and an index beyond the plan means the last.
 * Stages: FILL = before the piece is copied / read into its slot; LAUNCH = filled, the kernel launch fails; SYNC = the wait for
 * the piece's kernel fails (what a GPU dying under way looks like); DRAIN = the kernel finished, the failure comes before the
 * piece is copied back; AFTER_DRAIN = the piece HAS been copied back, then the failure.  stage < 0 disarms.
 * STALL is not a failure of a HIP call but of the HOST: the pipeline thread holds the piece back until the host-fed kernel has
 * given the call up by itself (its patience, MODGPU_TUNABLE_FEED_PATIENCE_MS, has run out and its stream has gone idle; bounded
 * at two minutes) and only then copies it in -- the call must end like any other that lost its GPU under way.  Who reaches the
 * piece first, kernel or pipeline, does not matter (only calls that take the host-fed kernel have this stage). */
enum { MODGPU_STAGE_FILL = 0, MODGPU_STAGE_LAUNCH = 1, MODGPU_STAGE_SYNC = 2, MODGPU_STAGE_DRAIN = 3, MODGPU_STAGE_AFTER_DRAIN = 4, MODGPU_STAGE_STALL = 5 };
#define MODGPU_INJECT_PIECE_LAST (-1)
#define MODGPU_INJECT_PIECE_MIDDLE (-2)
void modgpu_debug_inject_failure_at(int64_t piece, int stage);
/* 1 while an armed failure has not fired yet. */
int modgpu_debug_injection_armed(void);

/* forbid != 0: no staging set starts a worker thread from now on, as if thread creation failed (a pids / NPROC limit).  A call that
 * finds fewer workers than it has pipelines runs the rest itself, one after another -- and must not take a host-fed route, whose
 * pipelines have to run side by side. */
void modgpu_debug_forbid_worker_threads(int forbid);

/* Takes `count` pipeline slots of `device`'s staging set the way a large call does and keeps them until called again with 0 (returns
 * how many it holds).  With all 32 held, a header-sized call must still be served -- from the two slots only one-slot calls may take. */
int modgpu_debug_hold_slots(int device, int count);


#ifdef __cplusplus
}
#endif
#endif /* MODGPU_TESTING_H */
