/*
 * modgpu.h -- C ABI of the MI355X (gfx950) implementation of Modulate's cipher hot path.
 *
 * The reference (AdamClixby/Modulate) has no FFI or plugin interface; the seam it offers is
 * one C++ class,
 *
 *     class CEncryptionCycler { public: void Cycle(unsigned char*, unsigned int, int); ... };
 *                                                  (Modulate/CEncryptionCycler.h:3-10)
 *
 * called from exactly three places, always as  Cycle(buf + 4, size - 4, key):
 *     Modulate/CArk.cpp:338-339     CArk::Load            (header decrypt)
 *     Modulate/CArk.cpp:1135-1136   CArk::SaveArk         (header encrypt)
 *     Modulate/Modulate.cpp:485-486 Decode                (-decode command)
 *
 * This header is what that class's body binds to (modulate_amd/csrc/host/CEncryptionCycler.cpp is
 * the binding; INTEGRATION.md shows the same stub for the upstream tree).  Plain pointers and
 * sizes only; no C++ or torch types.  Every function returns MODGPU_OK (0) or a MODGPU_ERR_*
 * code, never throws, never prints; modgpu_last_error() gives the text for the calling thread.
 *
 * Semantics (bit-exact with Modulate/CEncryptionCycler.cpp:4-25):
 *     ks[i]  = low8( a^(i+1) * key mod (2^31-1) ) ^ 0xFF        a = 16807, residue 0 -> 2^31-1
 *     buf[j] ^= ks[stream_off + j]                               j = 0 .. n-1
 * stream_off = 0 reproduces one reference Cycle call.  n and stream_off are 64-bit, which lifts
 * the reference's `unsigned int` length cap (2^32-1) and lets one logical stream be split over
 * calls or devices.  Keys congruent to 0 mod 2^31-1 give the identity, as in the reference.
 *
 * Which engine computes.  Every modgpu_cycle_* / modgpu_hdr_* entry point below runs the gfx950
 * kernel and NOTHING ELSE: without a usable HIP device they fail with MODGPU_ERR_NO_DEVICE /
 * MODGPU_ERR_HIP.  The two exceptions are named for what they are:
 *     modgpu_cycle_scalar_host   the library's own host loop (never the GPU)
 *     modgpu_cycle_auto_host     the reference's "Cycle cannot fail" contract and its size dispatch: the
 *                                host loop for buffers below MODGPU_MIN_GPU_BYTES (header-sized: a kernel
 *                                launch costs more than the arithmetic) or when no GPU is usable, else the GPU
 * modgpu_path_stats() counts calls and bytes per engine, and MODGPU_REQUIRE_GPU=1 in the environment
 * forbids the host loop altogether (both entry points then fail with MODGPU_ERR_FORBIDDEN instead of
 * computing), so a test-suite or benchmark can prove which engine produced its bytes.
 *
 * Threading: callable concurrently from any number of host threads.  `device` selects the GPU
 * per call (-1 = the calling thread's current HIP device); no global "current device" is
 * relied on, and a call made with an explicit device leaves the calling thread's current HIP
 * device as it found it.  Host-buffer calls to the same device run side by side, each on staging slots of its own (a call
 * that finds too few free takes fewer pipelines; one that finds none waits for a release).  The library keeps parked worker
 * threads (for the staging pipelines per device and per NUMA node a caller's pages have been found on -- slots and workers sit
 * on the node of the pages they copy, only the GPU crosses the socket link; one pool for the host loop): started on first use,
 * never joined.
 *
 * Environment (each read once, when first needed) -- these ten and no others (tests/test_capi_cpu.py compares this list with the
 * strings of the built library):
 *     MODGPU_REQUIRE_GPU=1       no host loop anywhere (see above)
 *     MODGPU_MIN_GPU_BYTES=n     modgpu_cycle_auto_host's size threshold (default 16 MiB, the measured crossover
 *                                against one host thread; 0 = always the GPU)
 *     MODGPU_HOST_POLICY=name    what modgpu_cycle_auto_host does ABOVE that threshold when a GPU is usable:
 *                                offload (default) = the kernel -- the host's cores stay free and every GPU adds a link;
 *                                fastest = per call, the engine the committed crossover table prices as faster for this
 *                                size and memory kind, the host loop with the threads it would really get
 *     MODGPU_HOST_ISA=name       host-loop body: generic | avx2 | avx512 (default: the best the CPU runs)
 *     MODGPU_HOST_THREADS=n      most host threads one host-loop call may use (default min(cores, 32); never more than the
 *                                control group's CPU quota or the caller's affinity mask allow)
 *     MODGPU_HOST_PIPES=n        pipelines (host threads) one host-buffer call spreads its staging copies over (1..16, default 8)
 *     MODGPU_HOST_CHUNK_MB=n     largest page-locked staging slot in MiB (1..256, default 8): 2 x pipelines of them per caller at work
 *     MODGPU_DEVICE_ALIAS=n      see modgpu_device_count
 *     MODGPU_NUMA=0              do not place host memory and worker threads next to their GPU
 *     MODGPU_HELPER_BELOW_MHZ=n  shader clock below which the helper workgroups of a large launch join in (default: 77 % of the
 *                                device's peak shader clock, 1 848 MHz on MI355X; 0 = never)
 * (How a staged stream is cut and queued -- chunk count, ramp, lanes, copy flavour, ring depth -- and the host loop's binding and
 * control-group switches were environment variables until ABI 6.  They are constants now, the values the committed profiles
 * chose; the testing flavour of the library, libmodgpu_testing.so, still has them as knobs: include/modgpu_testing.h.)
 */
#ifndef MODGPU_H
#define MODGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MODGPU_OK 0
#define MODGPU_ERR_INVALID 1   /* bad argument (null pointer with n > 0, bad device index ...) */
#define MODGPU_ERR_NO_DEVICE 2 /* no HIP device visible                                        */
#define MODGPU_ERR_HIP 3       /* a HIP runtime call failed; see modgpu_last_error()           */
#define MODGPU_ERR_MAGIC 4     /* header magic is neither PS3 nor PS4 (eError_UnknownVersionNumber,
                                  Modulate/CArk.cpp:329-334, Modulate/Modulate.cpp:476-481)    */
#define MODGPU_ERR_IO 5        /* open / read / write of a part file failed (eError_FailedToOpenFile,
                                  eError_FailedToWriteData at Modulate/CArk.cpp:745-749, 883-889)     */
#define MODGPU_ERR_FORBIDDEN 6 /* the host loop was needed or asked for, and MODGPU_REQUIRE_GPU=1 forbids it */

/* Settings.h:16-20 */
#define MODGPU_MAGIC_PS3 0xc64eed30u
#define MODGPU_MAGIC_PS4 0x6f303f55u
#define MODGPU_KEY_PS3 0xc64eed30u
#define MODGPU_KEY_PS4 0x90cfc0abu

/* ABI version of this header (bumped on any signature change). */
#define MODGPU_ABI_VERSION 8
int modgpu_abi_version(void);

/* Number of HIP devices this library addresses (0 if none / runtime unusable).  Normally the
 * devices visible to the process; with MODGPU_DEVICE_ALIAS=N in the environment (a rehearsal
 * switch for multi-GPU code on a box with fewer GPUs) it is N logical devices, logical device d
 * running on physical device d mod <visible>, each with its own staging context and worker. */
int modgpu_device_count(void);

/* Text of the last error raised on the calling thread ("" if none).  Never NULL. */
const char *modgpu_last_error(void);

/* ---- the hot path ------------------------------------------------------------------ */

/* Replaces the loop body of CEncryptionCycler::Cycle (CEncryptionCycler.cpp:9-13) for a buffer
 * that is already device-resident.  `dev_buf` may have any byte alignment (the reference's
 * callers pass buf+4).  Asynchronous on `hip_stream` (a hipStream_t; NULL = the device's
 * null stream); the caller synchronises.  This is the entry point the roofline is measured on.
 * Allocation-free and capturable into a hipGraph.  Any number of EAGER launches may be in flight at once, on any streams,
 * beside any number of graph replays: the scheduling scratch of a large launch (a ticket counter) is never shared between
 * two launches that could overlap -- an eager launch gets scratch whose previous user has finished, or a launch shape that
 * needs none; a captured launch owns its scratch for good.
 * What the CALLER of a captured launch must ensure: the same captured node must not run twice at the same time.  That means
 * (a) do not launch one executable graph again -- on any stream -- while an earlier launch of it may still be running (CUDA
 * orders such launches itself; HIP does not document that it does, so this library does not rely on it), and (b) do not
 * launch two executable graphs instantiated from the same capture concurrently.  Both replay the same node, scratch
 * included: the tickets of the two runs would interleave and bytes would come out wrong WITHOUT an error.  Capture again
 * for every concurrent user.
 * Captured large launches draw their scratch from a grow-only pool of 1 024 lines per device that is never handed out
 * again (a graph may be replayed at any time).  A process that keeps re-capturing exhausts it; captures beyond that -- and a
 * capture that is the device's very first large launch -- take the static streaming shape, which needs no scratch and is
 * correct but ~7 % slower at 4 GiB.  modgpu_queue_stats (modgpu_testing.h) counts both. */
int modgpu_cycle_device(void *dev_buf, uint64_t n, int32_t key, uint64_t stream_off,
                        int device, void *hip_stream);

/* n_parts device-resident buffers of ONE device, each its own Cycle call under one key -- its own keystream, from
 * stream_offs[i], or from 0 when stream_offs is NULL: what a caller does with the parts of an archive (the reference
 * treats them as independent files, CArk.cpp:741-755, 849-897).  Asynchronous on hip_stream like
 * modgpu_cycle_device; buffers may have any alignment and any size (empty ones are skipped); they must not overlap.
 * Runs of up to 16 parts share ONE kernel launch when together they are beyond 256 MiB, or no more than 24 MiB each on
 * average: the fixed cost of a launch (~7 us: pipeline fill, the finishing spread, the gap to the next launch) is 5 % of
 * a 411 MB part and most of the time of a small one (measured: +5 % at 8 x 411 MB, +30 % at 16 x 50 MB, 3x at 16 x 1 MiB). */
int modgpu_cycle_batch_device(void *const *dev_parts, const uint64_t *sizes, const uint64_t *stream_offs, int n_parts,
                              int32_t key, int device, void *hip_stream);

/* Replaces CEncryptionCycler::Cycle (CEncryptionCycler.cpp:4-14) for a caller-owned HOST buffer,
 * on the GPU.  Pageable memory is staged through page-locked slots owned by this library (memcpy ->
 * slot -> kernel across PCIe on the slot -> memcpy back, chunked over several host threads and
 * overlapped; below 2 GiB ONE kernel serves the whole call and takes each chunk when its copy in has
 * landed); memory from modgpu_host_alloc / modgpu_host_register is cycled where it lies by one
 * kernel across PCIe, with no staging copy.  Synchronous: on return host_buf holds the result.
 * Never retains or frees host_buf. */
int modgpu_cycle_host(uint8_t *host_buf, uint64_t n, int32_t key, uint64_t stream_off, int device);

/* The library's own host loop for the same arithmetic (closed form of CEncryptionCycler.cpp:16-25,
 * sixteen independent byte states like one GPU lane-word; threads for large buffers).  This is
 * product code for hosts without a GPU -- it shares nothing with the test oracle under oracle/. */
int modgpu_cycle_scalar_host(uint8_t *host_buf, uint64_t n, int32_t key, uint64_t stream_off);

/* What CEncryptionCycler::Cycle binds to (`if (n < threshold || !gpu_ok) cpu_loop(); else ...`).
 * n < MODGPU_MIN_GPU_BYTES -- the headers the reference's three call sites pass -- is served by the host loop,
 * which finishes such a buffer before a kernel launch would have returned; larger buffers by modgpu_cycle_host.
 * The reference's Cycle returns void and cannot fail (CEncryptionCycler.cpp:4-14) and its callers do not guard it
 * (CArk.cpp:338-339, 1135-1136, Modulate.cpp:485-486), so the host loop finishes the call
 *   - when no GPU is visible, or the GPU attempt fails before it has changed host_buf: the whole buffer;
 *   - when the GPU is lost AFTER the call has begun on ordinary (pageable) memory -- what an unmodified caller passes,
 *     `new char[]` at CArk.cpp:320, 738, 780 -- : the buffer travels in pieces through page-locked slots and a piece changes
 *     host_buf only when it is copied back whole, so the library knows which pieces have arrived; the other pipelines of the
 *     call stop at once and the host loop does exactly the pieces that have not (keystream position = stream_off + the
 *     piece's offset).  modgpu_path_stats().midcall_rescues counts such calls.
 * ONE case stays an error: page-locked memory (modgpu_host_alloc / _register) is cycled where it lies by one kernel across
 * PCIe; if that kernel dies under way nobody knows which bytes it had written, and the plaintext exists nowhere else.  The call
 * then returns MODGPU_ERR_HIP with host_buf in an undefined state.  (A caller that must survive even that keeps its own copy,
 * or passes pageable memory.)  With MODGPU_REQUIRE_GPU=1 there is no second engine: every size runs on the kernel and a GPU
 * error is returned.
 * Above the threshold the default policy is to OFFLOAD: on a host with many cores the threaded host loop is faster than one
 * GPU's PCIe link for host-resident data (the link, ~50 GB/s, is the bound), but the kernel leaves those cores to the caller
 * and scales with the number of GPUs; MODGPU_HOST_POLICY=fastest picks the faster engine per call instead. */
int modgpu_cycle_auto_host(uint8_t *host_buf, uint64_t n, int32_t key, uint64_t stream_off, int device);

/* Header framing of CArk::Load (CArk.cpp:328-339) and Decode (Modulate.cpp:475-486):
 * LE u32 magic at hdr[0..3] selects the key, the cipher covers hdr[4..size).  Host buffer; the cipher call is
 * Cycle's, so the engine is chosen exactly as by modgpu_cycle_auto_host (headers are at most 512 KiB,
 * CArk.cpp:911-912: the host loop unless MODGPU_REQUIRE_GPU=1 or MODGPU_MIN_GPU_BYTES says otherwise).
 * An unknown magic is MODGPU_ERR_MAGIC before anything is touched. */
int modgpu_hdr_decrypt_host(uint8_t *hdr, uint64_t size, int device);

/* Header framing of SaveArk (CArk.cpp:914-915, 1135-1136): stores the platform magic at
 * hdr[0..3] (ps4 != 0 -> PS4) and encrypts hdr[4..size) with the platform key.  Host buffer; engine as above;
 * on failure the buffer is as it was. */
int modgpu_hdr_encrypt_host(uint8_t *hdr, uint64_t size, int ps4, int device);

/* Part-level sharding beside CArk::LoadArkData / lSaveArk (CArk.cpp:723-758, 845-899): part i
 * is an independent stream (its own Cycle from offset 0) and goes to GPU  i mod n_devices,
 * one host thread per GPU, no inter-GPU traffic.  n_devices <= 0 means all devices. */
int modgpu_cycle_parts_host(uint8_t *const *parts, const uint64_t *sizes, int n_parts,
                            int32_t key, int n_devices);

/* ONE host buffer over several GPUs: contiguous spans (multiples of 2 MiB, at least 64 MiB each), span d on GPU d with
 * stream offset stream_off + its position -- jump-ahead makes every span an independent stream, so there is still no
 * exchange step -- each through its own PCIe link, one host thread per GPU.  n_devices <= 0 means all devices;
 * a buffer under 128 MiB stays on one GPU.  Result identical to modgpu_cycle_host. */
int modgpu_cycle_host_split(uint8_t *host_buf, uint64_t n, int32_t key, uint64_t stream_off, int n_devices);

/* The same for parts that are already resident in HBM, part i on GPU devices[i] (BASELINE config 3: 8 x 4 GiB,
 * one per GPU).  The launches are asynchronous, so the calling thread alone keeps every GPU busy; the call
 * returns when all of them have finished.  No inter-GPU traffic.  Parts that share a GPU go to it through
 * modgpu_cycle_batch_device.
 * Ordering: the kernels run on a stream of the library's own per device (non-blocking).  Work the caller queued BEFORE the call
 * on a device's NULL stream or on any of its blocking streams -- an asynchronous upload or memset of a part, a kernel that
 * produces it -- is finished before that device's parts are cycled (the library's stream waits for an event recorded on the
 * NULL stream at entry).  Work on the caller's own NON-blocking streams is not ordered: synchronise those before the call. */
int modgpu_cycle_parts_device(void *const *dev_parts, const uint64_t *sizes, const int *devices, int n_parts, int32_t key);

/* ---- part files streamed through the GPU ----------------------------
 * The reference reads a part with one fread into the concatenated buffer (CArk.cpp:751) and writes
 * a slice with one fwrite (CArk.cpp:883).  These do the same transfers with the cipher applied on
 * the way, overlapped, several chunks in flight: pread into a page-locked slot -> the kernel cycles the
 * slot across PCIe (file -> memory below 2 GiB: ONE kernel launch for the whole call) -> copy / pwrite
 * out; page-locked caller memory on the SOURCE side is DMA'd from where it lies.  Each call is one
 * stream whose first byte has keystream position stream_off (0 = a part's own Cycle). */

/* Whole file src_path -> dst_path (created / truncated).  The two may name the same file, by any
 * spelling (compared by device and inode): it is then cycled in place.
 * All three file routes share one rule for a GPU that is LOST AFTER THE CALL HAS BEGUN: the source still holds every byte (a
 * destination file is written piece by piece, each only when it is finished, also in place), so the pieces that have not
 * arrived are read again and done by the library's host loop -- unless MODGPU_REQUIRE_GPU=1; modgpu_path_stats().midcall_rescues
 * counts such calls.  Without a usable GPU at the start they fail like every other kernel entry point. */
int modgpu_cycle_file(const char *src_path, const char *dst_path, int32_t key, uint64_t stream_off, int device);

/* n bytes at byte offset file_off of `path` -> host_dst[0..n) (any kind of memory; the rule above holds whatever host_dst is). */
int modgpu_cycle_file_to_host(const char *path, uint64_t file_off, uint8_t *host_dst, uint64_t n, int32_t key,
                              uint64_t stream_off, int device);

/* host_src[0..n) -> `path` (created / truncated).  host_src is not modified. */
int modgpu_cycle_host_to_file(const uint8_t *host_src, uint64_t n, const char *path, int32_t key, uint64_t stream_off,
                              int device);

/* ---- page-locked host memory -----------------------------------------------------------------
 * Replaces the `new char[total]` of CArk::LoadArkData / BuildArk (CArk.cpp:738, 780) for callers
 * that will cycle the buffer: the GPU's DMA engines and kernels reach these pages directly, so
 * the host-buffer entry points above skip both staging copies for any range inside them.
 * Without a GPU, or if the pages cannot be locked (locked-memory limit), the memory is ordinary (64-byte
 * aligned) and everything still works through the staged route; modgpu_host_is_pinned tells which it is. */
int modgpu_host_alloc(void **host_ptr, uint64_t n);
int modgpu_host_free(void *host_ptr);
/* The same, placed for a multi-socket node: the pages are bound to the NUMA node the GPU hangs off
 * (/sys/bus/pci/devices/<bdf>/numa_node; mbind, preferred policy) before they are locked, so the bytes that
 * cross PCIe to that GPU come from its own socket's DRAM.  modgpu_host_alloc_parts makes ONE contiguous buffer
 * for a list of parts laid end to end -- the concatenated buffer of CArk::LoadArkData / BuildArk (CArk.cpp:738,
 * 780) -- with part i's pages next to GPU i mod n_devices, the GPU modgpu_cycle_parts_host sends it to
 * (n_devices <= 0: all).  With MODGPU_NUMA=0 or without a GPU both are modgpu_host_alloc.  Otherwise
 * modgpu_host_alloc_parts always takes its own route -- reserve, bind each part's pages where its GPU's node is known (a part
 * whose GPU's node cannot be read, e.g. no numa_node in a container's sysfs, is simply not bound), first-touch from several
 * threads, page-lock in place -- because that route is also the faster way to get GB-sized page-locked memory (0.27 s against
 * 0.43-0.51 s for 3.3 GB); modgpu_host_alloc_near falls back to modgpu_host_alloc when its one device's node is unknown.
 * Free with modgpu_host_free.  Worker threads of the host-buffer routes run on their GPU's node as well.  Placement is
 * best effort and changes no result. */
int modgpu_host_alloc_near(void **host_ptr, uint64_t n, int device);
int modgpu_host_alloc_parts(void **host_ptr, const uint64_t *sizes, int n_parts, int n_devices);
/* NUMA node of the GPU behind `device`, -1 if unknown or placement is off. */
int modgpu_device_numa_node(int device);
/* For callers that cannot change how their buffer is allocated: page-locks [host_ptr, host_ptr + n) where it
 * lies (hipHostRegister) so that later cycles of ranges inside it take the no-copy route.  Pinning costs
 * about as much as one staged pass over the buffer, so it pays from the second cycle on.  Unregister
 * before freeing the memory.  Without a GPU both calls succeed and do nothing. */
int modgpu_host_register(void *host_ptr, uint64_t n);
int modgpu_host_unregister(void *host_ptr);
/* 1 if [p, p+n) lies inside one page-locked, device-visible allocation or registration, else 0. */
int modgpu_host_is_pinned(const void *p, uint64_t n);

/* ---- which engine ran ---------------------------------------------------------------------- */
typedef struct modgpu_path_stats {
    uint64_t gpu_calls;      /* host-buffer / file calls served by the kernel                    */
    uint64_t gpu_bytes;      /* payload bytes those calls cycled                                 */
    uint64_t gpu_launches;   /* kernel launches, modgpu_cycle_device included                    */
    uint64_t scalar_calls;   /* calls served by the host loop (direct or through _auto_)         */
    uint64_t scalar_bytes;
    uint64_t staged_bytes;   /* of gpu_bytes: went through a pageable<->pinned memcpy            */
    uint64_t direct_bytes;   /* of gpu_bytes: DMA'd or read straight from the caller's pinned pages */
    uint64_t auto_fallbacks; /* modgpu_cycle_auto_host calls that ended on the host loop because the GPU could not serve them */
    uint64_t auto_small;     /* modgpu_cycle_auto_host calls served by the host loop because n < MODGPU_MIN_GPU_BYTES */
    uint64_t auto_policy_host; /* modgpu_cycle_auto_host calls of n >= MODGPU_MIN_GPU_BYTES that MODGPU_HOST_POLICY=fastest kept on the host loop */
    uint64_t midcall_rescues;       /* calls (modgpu_cycle_auto_host, the modgpu_cycle_file* routes) whose GPU was lost AFTER the call had begun and
                                       that the host loop finished: counted in gpu_calls AND -- for _auto_ -- in auto_fallbacks */
    uint64_t midcall_rescued_bytes; /* bytes of those calls the host loop did (in scalar_bytes, not in gpu_bytes) */
} modgpu_path_stats_t;
/* Process-wide counters since load (or the last reset).  reset != 0 zeroes them after the read. */
int modgpu_path_stats(modgpu_path_stats_t *out, int reset);
/* 1 if MODGPU_REQUIRE_GPU=1 was set when the library was loaded. */
int modgpu_gpu_required(void);
/* modgpu_cycle_auto_host's size threshold as latched from MODGPU_MIN_GPU_BYTES. */
uint64_t modgpu_min_gpu_bytes(void);
/* "offload" or "fastest": MODGPU_HOST_POLICY as latched.  Static storage. */
const char *modgpu_host_policy(void);
/* What `fastest` would decide for one call over n bytes of pageable (pinned = 0) or page-locked (1) memory on this host:
 * returns 1 for the host loop, 0 for the kernel, and the two priced durations in microseconds (either pointer may be NULL).
 * The prices come from the crossover table the library was built with (modulate_amd/csrc/crossover_table.h). */
int modgpu_host_policy_engine(uint64_t n, int pinned, double *host_us, double *kernel_us);
/* The host-loop body this process uses: "generic", "avx2" or "avx512".  Static storage. */
const char *modgpu_host_loop_isa(void);

/* ---- device-memory helpers (bench / tests / callers that keep parts resident) ---------
 * hipMalloc / hipFree / hipMemcpy / a synchronize on the named device -- plus two pieces of housekeeping that are NOT in their
 * names, both best effort (whatever fails there is retried or simply paid by the caller's first launch) and both outside
 * anybody's timed launch:
 *   modgpu_alloc   the FIRST allocation on a device in this process also prepares the device: the code object is loaded, the
 *                  work-queue kernel's ticket ring is set up, and two real, tiny work-queue launches (2 x 64 KiB of scratch) run
 *                  and are waited for on a private stream.  A process's first real launch of that kernel otherwise costs 10 ms
 *                  (code object) + 15-35 us (kernel function, ring) inside whatever the caller times
 *                  (profiles/r03_first_pass.txt, r05_first_launch.txt).
 *   modgpu_h2d     a copy of 1 MiB or more first enqueues ONE EMPTY KERNEL (one workgroup, no words) on a private non-blocking
 *                  stream that nobody waits for: an upload keeps only the DMA engines busy, the shader engines fall asleep within a
 *                  fraction of a second, and the first launch behind the upload would pay their wake-up (~15 us on a 411 MB part:
 *                  0.72 -> 0.81 of the HBM peak for that launch).  Costs the caller one asynchronous launch call per copy.
 * A caller that brings its own device memory (hipMalloc / hipMemcpy of its own, a torch tensor) gets neither -- unless it says so:
 *   modgpu_prepare(device)   both of the above by name: prepares the device if this process has not yet, and enqueues the empty
 *                  wake-up launch.  Call it when your own upload STARTS (or any time before the first modgpu_cycle_device).
 *                  Never required for correctness.  Returns MODGPU_OK, or the error of selecting the device. */
int modgpu_alloc(void **dev_ptr, uint64_t n, int device);
int modgpu_free(void *dev_ptr, int device);
int modgpu_h2d(void *dev_dst, const void *host_src, uint64_t n, int device);
int modgpu_d2h(void *host_dst, const void *dev_src, uint64_t n, int device);
int modgpu_sync(int device, void *hip_stream);
int modgpu_prepare(int device);

/* ---- host-side jump-ahead arithmetic (exposed so it can be checked without a GPU) ---- */

/* State the reference loop holds when it XORs stream byte i: a^(i+1)*key mod m, in [1, m];
 * this is what the library feeds the kernel as its per-launch base. */
uint32_t modgpu_state_at(int32_t key, uint64_t i);

/* Fills out[0..count) with the kernel's compile-time jump tables so tests can verify them
 * against independent arithmetic.  which: 0 = a^j (j<16), 1 = a^(16*t) (t<256),
 * 2 = a^(4096*b) (b<256), 3 = a^(4096*256*b) (b<256).  Returns entries written. */
int modgpu_jump_table(int which, uint32_t *out, int count);

#ifdef __cplusplus
}
#endif
#endif /* MODGPU_H */
