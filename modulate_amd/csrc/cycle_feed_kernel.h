// cycle_feed_kernel.h -- launch interface of the HOST-FED kernel: ONE launch per host-buffer call on the pageable (staged)
// route (host_stream.cpp), instead of one launch per chunk.
//
// The staged route copies a caller's pageable bytes chunk by chunk into page-locked slots, has the GPU cycle each slot across
// PCIe where it lies, and copies the result out.  With a kernel launch per chunk the link carries ~64 short kernels per call,
// 2-3 at once, each with its own ramp-up and drain: 45 GB/s between them where one long kernel holds 50
// (profiles/r05_pcie_route_staged_64MiB.json).  This kernel is launched when the call starts and stays: a workgroup draws a
// ticket -- a 32 KiB piece of the stream, in stream order --, waits until the host has marked the piece's chunk `ready`
// (a word in page-locked host memory), cycles the piece in its slot, and counts it; the workgroup that completes a chunk marks
// it `done`, which the host thread that owns the chunk polls before it copies the chunk out.  The link never idles between
// chunks as long as the host's copies keep ahead (profiles/r05_pcie_persist.txt: lab form, +11 % at 16 MiB, +7 % at 64 MiB).
//
// A kernel that waits for the host must end whatever the host does: every wait gives up when the host raises `abort`
// (a pipeline failed, host_stream.cpp) or after `patience_ticks` of the 100 MHz wall clock without the chunk turning up; a
// workgroup that gives up says so in work[1] and leaves -- and so does every other workgroup the next time it would have to
// wait, since work[1] != 0 ends a wait like `abort` does --, and the tickets it would have drawn are never done -- the host sees
// chunks without a `done` mark after the kernel has ended and treats the call as lost mid-way (finish on the host loop, or the
// error, as for any other GPU failure).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

constexpr int CYCLE_FEED = 4; // modgpu_last_launch's `variant` for this kernel (reporting only; cycle_kernel.h's CycleVariant ends at 3)
constexpr int kFeedSlotsMax = 32;             // 16 pipelines x 2 slots
constexpr uint32_t kFeedPieceBytes = 32768u;  // a ticket: 8 trips of 256 lanes x 16 bytes
constexpr uint32_t kFeedPiecesMax = 1u << 24; // the three-byte jump tables: 512 GiB per call
struct CycleFeedArgs {
    uint8_t *slot[kFeedSlotsMax]; // the call's slots as the device addresses them: chunk k is in slot[(k % pipes) * 2 + (k / pipes) % 2]
    const uint32_t *ready;        // [chunks] host memory, written by the host: chunk k has been copied into its slot
    const uint32_t *abort;        // host memory: the call is lost, leave
    uint32_t *done;               // [chunks] host memory, written by the kernel: chunk k's slot holds the result
    uint32_t *work;               // device memory, zero at launch: [0] ticket counter, [1] workgroups that gave up, [2 + k] pieces of chunk k finished
    uint64_t n;                   // bytes of the call (the last piece may be short and need not end on a 16-byte word)
    uint64_t patience_ticks;      // longest wait for one chunk, in ticks of the 100 MHz wall clock
    uint32_t chunk_bytes;         // a multiple of kFeedPieceBytes, at most the slot size
    uint32_t pipes;               // pipelines of the call (each owns two slots)
    uint32_t base;                // canonical state of the call's first byte
};
uint32_t modgpu_feed_block();
const char *modgpu_feed_kernel_name();
hipError_t modgpu_launch_cycle_feed(const CycleFeedArgs &a, uint32_t grid, hipStream_t stream);
