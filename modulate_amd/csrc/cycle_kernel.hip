// cycle_kernel.hip -- the gfx950 (CDNA4) form of CEncryptionCycler::Cycle's inner loop
// (Modulate/CEncryptionCycler.cpp:9-13).  Device code: cycle_kernel_impl.h.
//
// The reference walks ONE Park-Miller state through the buffer, one step per byte.  Here every
// lane owns independent, 16-byte-aligned, coalesced HBM words (one buffer_load_dwordx4 /
// buffer_store_dwordx4 per word; 64 lanes = 1 KiB contiguous, line-aligned, per wave instruction)
// and reaches its keystream position directly:
//
//   state of a word's first byte     S   = base * a^(4096*tile) * a^(16*(tid%256))   (jump tables)
//   state of byte j of the word      S_j = S * a^j mod m,  j = 1..15                  (independent)
//   keystream byte                   ks  = low8(S_j) ^ 0xFF
//
// Integer only: this is HBM-bound byte work, no MFMA; LDS holds nothing but an 8-byte ticket mailbox.  One byte costs
// two v_mad_u64_u32 (the product, then the Mersenne fold 2^31 == 1 mod m as a second multiply-add, whose carry-out IS
// the canonicalising +1) and one SDWA add-with-carry that packs it: 20 issue cycles; "^0xFF" and the data XOR are one
// v_xnor per dword.  (ALG 2 in cycle_kernel_impl.h; the small shape keeps ALG 1, one instruction more per byte.)
//
// Three launch shapes (cycle_kernel.h):
//   queue   what the roofline is measured on (buffers > 256 MiB): persistent 1024-thread workgroups, 25 per 32 CUs,
//           64 KiB chunks on absolute 64 KiB-aligned addresses handed out by a ticket counter (a static prefix of
//           two, then tickets fetched at the start of the trip before the one that loads them), so fast and slow XCDs finish together; the next chunk's
//           four loads are in flight while this one is computed; loads and stores are issued as
//           workgroup-synchronous bursts (nt loads, sc1+nt stores).  The {ticket, done} pair is cleaned by the last
//           workgroup out, which then signs off in a host-visible word so the host never hands a pair to two
//           launches that could overlap (modgpu_capi.cpp: queue_pair).
//   large   the same bursts with the static chunk map (b, b+G, ...), 128 KiB chunks, one workgroup per CU: what a
//           launch takes when no ticket pair is free, or beyond 2^24 chunks.
//   small   256 threads x one word, no pipeline: headers ... 256 MiB, and page-locked host memory across PCIe.
// The queue shape takes a TABLE of buffers (CycleQueueArgs; one buffer = a table of one): an archive's parts resident on one
// GPU share one launch and pay its fixed cost once; the table travels in the kernel arguments.
// DESIGN.md 3-4 has the measurements behind each of these choices.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "cycle_kernel_impl.h"

#include <cstdio>

namespace {
// One launch shape = one instantiation; everything the host layer asks about a shape comes from here.
template <int U, int BLOCK, int ALG, bool STREAM> struct Shape {
    static constexpr uint32_t chunk = (uint32_t)U * BLOCK * lcg::WORD;
    static constexpr uint32_t block = BLOCK;
    static void launch(const CycleArgs &a, uint32_t grid, hipStream_t stream)
    {
        hipLaunchKernelGGL((modgpu_cycle_kernel<U, BLOCK, ALG, STREAM>), dim3(grid), dim3(BLOCK), 0, stream, a);
    }
    static const char *name() // as a profiler prints it
    {
        static char buf[96];
        static const int n = std::snprintf(buf, sizeof buf, "modgpu_cycle_kernel<%d, %d, %d, %s>", U, BLOCK, ALG, STREAM ? "true" : "false");
        (void)n;
        return buf;
    }
};
// the streaming shape fed from a ticket counter instead of the static chunk map
template <int U, int BLOCK> struct QueueShape {
    static constexpr uint32_t chunk = (uint32_t)U * BLOCK * lcg::WORD;
    static constexpr uint32_t block = BLOCK;
    static void launch(const CycleQueueArgs &a, uint32_t grid, hipStream_t stream)
    {
        hipLaunchKernelGGL((modgpu_cycle_queue_kernel<U, BLOCK>), dim3(grid), dim3(BLOCK), 0, stream, a);
    }
    static const char *name()
    {
        static char buf[96];
        static const int n = std::snprintf(buf, sizeof buf, "modgpu_cycle_queue_kernel<%d, %d>", U, BLOCK);
        (void)n;
        return buf;
    }
};
// one word per thread, 256 threads, no pipeline: launch-latency-bound sizes (the 4-instruction keystream sequence)
using Small = Shape<1, 256, 1, false>;
// U=8 words x 1024 threads, three-instruction keystream, pipelined + loads-first, sc1 stores, workgroup-synchronous bursts
using Large = Shape<8, 1024, 2, true>;
// 1024 threads x 4 words = 64 KiB chunks: measured best under the queue (profiles/r02_tune_cycle_queue_shapes.txt,
// every row validated): 128 KiB chunks balance coarser, 32 KiB and below saturate the ticket counter (~80 tickets/us
// chip-wide), 512-thread workgroups and a second chunk of loads in flight lose 1 %; stores sc1+nt gain 0.5-1.8 %
// over sc1 alone at every size
using Queue = QueueShape<4, 1024>;
} // namespace

uint32_t modgpu_queue_chunk_bytes() { return Queue::chunk; }
uint32_t modgpu_queue_block() { return Queue::block; }
const char *modgpu_queue_kernel_name() { return Queue::name(); }
hipError_t modgpu_launch_cycle_queue(const CycleQueueArgs &a, uint32_t grid, hipStream_t stream)
{
    Queue::launch(a, grid, stream);
    return hipGetLastError();
}

uint32_t modgpu_variant_chunk_bytes(int variant)
{
    return variant == CYCLE_QUEUE ? Queue::chunk : variant == CYCLE_LARGE ? Large::chunk : Small::chunk;
}
uint32_t modgpu_variant_block(int variant)
{
    return variant == CYCLE_QUEUE ? Queue::block : variant == CYCLE_LARGE ? Large::block : Small::block;
}
const char *modgpu_variant_kernel_name(int variant)
{
    return variant == CYCLE_QUEUE ? Queue::name() : variant == CYCLE_LARGE ? Large::name() : Small::name();
}

hipError_t modgpu_launch_cycle(const CycleArgs &a, int variant, uint32_t grid, hipStream_t stream)
{
    if (variant == CYCLE_LARGE) Large::launch(a, grid, stream);
    else Small::launch(a, grid, stream);
    return hipGetLastError();
}
