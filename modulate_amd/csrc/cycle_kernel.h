// cycle_kernel.h -- launch interface between the host layer (modgpu_capi.cpp) and the kernel TU.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// Everything the kernel needs, precomputed on the host (modgpu_capi.cpp: plan_cycle()).
// All `base_*` values are canonical states in [1, 2^31-2] (key != 0 mod m is checked earlier).
struct CycleArgs {
    void *body;          // 16-byte aligned start of the body
    uint64_t body_words; // full 16-byte words in the body
    uint32_t base_body;  // state of the byte `lead` positions BEFORE the body's first byte (see lead)
    uint32_t stride_mul2; // 2 * a^(chunk_bytes * gridDim): advances a lane-word by one grid trip (pre-doubled
                          // for the kernel's split-at-bit-32 multiply, see mul_fold)
    uint8_t *head_ptr;   // first byte of the buffer (head_n < 16 bytes before the body)
    uint8_t *tail_ptr;   // first byte after the body (tail_n < 16 bytes)
    uint32_t head_n, tail_n;
    uint32_t lead;       // body address modulo the variant's chunk size: chunks sit on absolute chunk-aligned
                         // addresses, so the first one starts `lead` bytes before the body and is masked there
    uint32_t base_head;  // state of the buffer's first byte
    uint32_t base_tail;  // state of the first tail byte
};

// ---- the work-queue shape: ONE launch over one or several buffers ---------------------------------------------------
// Chunks are handed out by a ticket counter, so the kernel does not care whose chunk it is: the parts' chunks form one
// index space -- part p owns [start[p], start[p+1]) -- and every part is its own keystream (its own base states), has
// its own alignment (lead) and its own ragged edges.  A single buffer is a table of one part.  Several parts in a launch
// pay the launch's fixed cost (~7 us: pipeline fill, finishing spread, gap to the next launch) once: that cost is 5 % of
// a 411 MB part (BASELINE config 4's part size).  The table travels in the kernel arguments.
constexpr int kCycleBatchMax = 16;
struct CycleQueuePart {
    uint8_t *body;      // 16-byte aligned start of the part's body
    uint64_t end;       // lead + body bytes: one past the body's last byte, counted from the chunk origin (body - lead)
    uint32_t lead;      // body address modulo the chunk size (the cut first chunk, if any, is not in the index space:
                        // workgroup p does it for part p, with the part's edges)
    uint32_t base_body; // state of the byte at the chunk origin
    uint32_t base_head, base_tail;
    uint32_t head_n, tail_n;
};
struct CycleQueueArgs {
    uint32_t *queue;      // {ticket counter, workgroups done}, both 0 at launch and 0 again at exit
    uint32_t *queue_done; // host-visible word that receives queue_seq once the pair is clean again (the host hands a pair
    uint32_t queue_seq;   // out again only after it has seen that); nullptr: nobody waits for this pair
    uint32_t main_groups;      // workgroups [0, main_groups) stream from the start; the rest are HELPERS, which measure the
    uint32_t helper_below_mhz; // shader clock when they start and join (tickets only) while it is below this many MHz, else
                               // leave at once.  0 main_groups = every workgroup is a main one.
    uint32_t n_parts;                   // 1 .. kCycleBatchMax
    uint32_t start[kCycleBatchMax + 1]; // first global chunk index of each part; start[n_parts] = total; unused entries = total
    CycleQueuePart part[kCycleBatchMax];
};
uint32_t modgpu_queue_chunk_bytes();
uint32_t modgpu_queue_block();
const char *modgpu_queue_kernel_name();
hipError_t modgpu_launch_cycle_queue(const CycleQueueArgs &a, uint32_t grid, hipStream_t stream);

// Launch shapes.  A workgroup trip covers `chunk_bytes` contiguous bytes; the grid strides over
// chunks.  grid * chunk_bytes / 4096 must stay <= 65536 (two-level tile jump table).
enum CycleVariant : int {
    CYCLE_SMALL = 0, // 256 threads x 1 word : 4 KiB chunks, headers and other small buffers
    CYCLE_LARGE = 1, // 1024 threads x 8 words, software-pipelined, workgroup-synchronous bursts: 128 KiB chunks
    CYCLE_QUEUE = 2, // persistent workgroups, 64 KiB chunks handed out by a ticket counter (CycleQueueArgs; < 2^24 chunks per part)
    CYCLE_BATCH = 3, // reporting only (modgpu_last_launch): a work-queue launch that carried several parts
};
constexpr int kCycleVariants = 3;
uint32_t modgpu_variant_chunk_bytes(int variant);
uint32_t modgpu_variant_block(int variant);
// The instantiation's name as a profiler prints it ("modgpu_cycle_kernel<8, 1024, 2, true>"),
// generated from the same template arguments the launch uses, so it cannot go stale.
const char *modgpu_variant_kernel_name(int variant);

// CYCLE_SMALL or CYCLE_LARGE (the work-queue shape takes a table: modgpu_launch_cycle_queue).  Returns hipGetLastError().
hipError_t modgpu_launch_cycle(const CycleArgs &a, int variant, uint32_t grid, hipStream_t stream);
