// cycle_kernel.hip -- the gfx950 (CDNA4) form of CEncryptionCycler::Cycle's inner loop
// (Modulate/CEncryptionCycler.cpp:9-13).
//
// The reference walks ONE Park-Miller state through the buffer, one step per byte.  Here every
// lane owns an independent, 16-byte-aligned, coalesced HBM word (one global_load_dwordx4 /
// global_store_dwordx4 per word; 64 lanes = 1 KiB contiguous per wave instruction) and reaches
// its keystream position directly:
//
//   state of the word's first byte   S   = base * a^(4096*tile) * a^(16*tid)      (jump tables)
//   state of byte j of the word      S_j = S * a^j mod m,  j = 1..15               (independent)
//   keystream byte                   ks  = low8(S_j) ^ 0xFF
//
// Integer only: this is HBM-bound byte work, no MFMA.  All multiplies are 32x32->64 with one
// Mersenne fold (2^31 == 1 mod m); the "^0xFF" and the data XOR collapse into one XNOR per dword.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "cycle_kernel_impl.h"

uint32_t modgpu_variant_chunk_bytes(int variant)
{
    return variant == CYCLE_LARGE ? 8u * 1024u * lcg::WORD : 1u * 256u * lcg::WORD;
}

hipError_t modgpu_launch_cycle(const CycleArgs &a, int variant, uint32_t grid, hipStream_t stream)
{
    if (variant == CYCLE_LARGE)
        hipLaunchKernelGGL((modgpu_cycle_kernel<8, 1024, 1, 2, MODE_FULL, AUX_SC1, 3>), dim3(grid), dim3(1024), 0, stream, a);
    else
        hipLaunchKernelGGL((modgpu_cycle_kernel<1, 256, 1, 0, MODE_FULL>), dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}
