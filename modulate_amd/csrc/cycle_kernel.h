// cycle_kernel.h -- launch interface between the host layer (modgpu_capi.cpp) and the kernel TU.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// Everything the kernel needs, precomputed on the host (modgpu_capi.cpp: plan_cycle()).
// All `base_*` values are canonical states in [1, 2^31-2] (key != 0 mod m is checked earlier).
struct CycleArgs {
    void *body;          // 16-byte aligned start of the body
    uint64_t body_words; // full 16-byte words in the body
    uint32_t base_body;  // state of the body's first byte
    uint32_t stride_mul; // a^(4096 * U * gridDim): advances a lane-word by one grid trip
    uint8_t *head_ptr;   // first byte of the buffer (head_n < 16 bytes before the body)
    uint8_t *tail_ptr;   // first byte after the body (tail_n < 16 bytes)
    uint32_t head_n, tail_n;
    uint32_t base_head;  // state of the buffer's first byte
    uint32_t base_tail;  // state of the first tail byte
};

// grid * unroll must be <= 65536 (two-level tile table).  Returns hipGetLastError().
hipError_t modgpu_launch_cycle(const CycleArgs &a, int unroll, uint32_t grid, hipStream_t stream);
