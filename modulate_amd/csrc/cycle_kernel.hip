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
// Integer only: this is HBM-bound byte work, no MFMA, no LDS.  One byte costs two v_mad_u64_u32 (the
// product, then the Mersenne fold 2^31 == 1 mod m as a second multiply-add), a shift and one SDWA
// add that canonicalises and packs it; "^0xFF" and the data XOR are one v_xnor per dword.
//
// Two launch shapes (cycle_kernel.h).  The streaming one is what the roofline is measured on:
// one persistent 1024-thread workgroup per CU, 128 KiB chunks on absolute 128 KiB-aligned
// addresses, the next chunk's eight loads in flight while this one is computed, loads and stores
// issued as workgroup-synchronous bursts (nt loads, sc1 stores).  DESIGN.md 3-4 has the measurements
// behind each of these choices.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "cycle_kernel_impl.h"

uint32_t modgpu_variant_chunk_bytes(int variant)
{
    return variant == CYCLE_LARGE ? 8u * 1024u * lcg::WORD : 1u * 256u * lcg::WORD;
}

hipError_t modgpu_launch_cycle(const CycleArgs &a, int variant, uint32_t grid, hipStream_t stream)
{
    if (variant == CYCLE_LARGE) // U=8 words x 1024 threads, SDWA keystream, pipelined + loads-first, sc1 stores, sync bursts
        hipLaunchKernelGGL((modgpu_cycle_kernel<8, 1024, 1, 2, MODE_FULL, AUX_SC1, 3>), dim3(grid), dim3(1024), 0, stream, a);
    else // one word per thread, 256 threads, no pipeline: launch-latency-bound sizes
        hipLaunchKernelGGL((modgpu_cycle_kernel<1, 256, 1, 0, MODE_FULL>), dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}
