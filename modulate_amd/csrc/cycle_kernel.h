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
    uint32_t *queue;     // work-queue shape only: {ticket counter, workgroups done}, both 0 at launch and 0 again at exit
    uint32_t *queue_done; // work-queue shape: host-visible word that receives queue_seq once the pair is clean again (the host
    uint32_t queue_seq;   // hands a pair out again only after it has seen that); nullptr: nobody waits for this pair
    uint32_t main_groups;      // work-queue shape: workgroups [0, main_groups) stream from the start; the rest are HELPERS, which
    uint32_t helper_below_mhz; // measure the shader clock when they start and join (tickets only) while it is below this many MHz,
                               // else leave at once.  0 main_groups = every workgroup is a main one.
    uint64_t *trace;     // nullptr in the product.  tools/tune_cycle's TRACE instantiation writes per-workgroup
                         // timestamps here (wall_clock64, 100 MHz): [blk*32+0] start, [+1+k] end of trip k, [+31] XCC id
};

// ---- several buffers in ONE launch of the work-queue shape (modgpu_cycle_batch_device) ------------------------------
// A launch costs ~7 us of pipeline fill, finishing spread and gap to the next one: 5 % of a 411 MB part (BASELINE
// config 4's part size).  Here the parts' chunks form one index space -- part p owns [start[p], start[p+1]) -- that the
// same ticket counter hands out, so the fixed cost is paid once per batch.  Every part is its own keystream (its own
// base states), has its own alignment (lead) and its own ragged edges.
constexpr int kCycleBatchMax = 16;
struct CycleBatchPart {
    uint8_t *body;      // 16-byte aligned start of the part's body
    uint64_t end;       // lead + body bytes: one past the body's last byte, counted from the chunk origin (body - lead)
    uint32_t lead;      // body address modulo the chunk size (the cut first chunk, if any, is not in the index space:
                        // workgroup p does it for part p, with the part's edges)
    uint32_t base_body; // state of the byte at the chunk origin
    uint32_t base_head, base_tail;
    uint32_t head_n, tail_n;
};
struct CycleBatchArgs {
    uint32_t *queue, *queue_done; // as in CycleArgs
    uint32_t queue_seq, main_groups, helper_below_mhz;
    uint32_t n_parts;                   // 1 .. kCycleBatchMax
    uint32_t start[kCycleBatchMax + 1]; // first global chunk index of each part; start[n_parts] = total; unused entries = total
    CycleBatchPart part[kCycleBatchMax];
};
uint32_t modgpu_batch_chunk_bytes();
uint32_t modgpu_batch_block();
const char *modgpu_batch_kernel_name();
hipError_t modgpu_launch_cycle_batch(const CycleBatchArgs &a, uint32_t grid, hipStream_t stream);

// Launch shapes.  A workgroup trip covers `chunk_bytes` contiguous bytes; the grid strides over
// chunks.  grid * chunk_bytes / 4096 must stay <= 65536 (two-level tile jump table).
enum CycleVariant : int {
    CYCLE_SMALL = 0, // 256 threads x 1 word : 4 KiB chunks, headers and other small buffers
    CYCLE_LARGE = 1, // 1024 threads x 8 words, software-pipelined, workgroup-synchronous bursts: 128 KiB chunks
    CYCLE_QUEUE = 2, // the same shape, chunks handed out by a ticket counter (needs CycleArgs::queue; < 2^24 chunks)
    CYCLE_BATCH = 3, // reporting only (modgpu_last_launch): the work-queue shape over several parts, CycleBatchArgs
};
constexpr int kCycleVariants = 3;
uint32_t modgpu_variant_chunk_bytes(int variant);
uint32_t modgpu_variant_block(int variant);
// The instantiation's name as a profiler prints it ("modgpu_cycle_kernel<8, 1024, 1, 2, 0, 16, 3>"),
// generated from the same template arguments the launch uses, so it cannot go stale.
const char *modgpu_variant_kernel_name(int variant);

// Returns hipGetLastError().
hipError_t modgpu_launch_cycle(const CycleArgs &a, int variant, uint32_t grid, hipStream_t stream);
