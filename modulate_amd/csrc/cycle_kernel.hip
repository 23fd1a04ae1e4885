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
#include "lcg.h"
#include "cycle_kernel.h"

namespace {

using u32x4 = uint32_t __attribute__((ext_vector_type(4))); // one dwordx4 lane-word

// Jump tables, baked into the code object (see lcg.h).  kTile* are read with block-uniform
// indices (scalar loads); kLanePow once per thread at start-up.
__constant__ lcg::Table<256> c_lane_pow = lcg::kLanePow;
__constant__ lcg::Table<256> c_tile_lo = lcg::kTileLo;
__constant__ lcg::Table<256> c_tile_hi = lcg::kTileHi;

// x, y canonical residues (< 2^31).  Returns x*y mod m, canonical (never 0 for non-zero inputs).
__device__ __forceinline__ uint32_t mulmod_canon(uint32_t x, uint32_t y)
{
    uint64_t p = (uint64_t)x * y;
    uint32_t X = ((uint32_t)p & lcg::M) + (uint32_t)(p >> 31); // < 2^32, == p (mod m)
    return (X & lcg::M) + (X >> 31);
}

// Low byte (bits 7..0, upper bits garbage-free) of canonical( s * a^J ), s canonical.
// The constant is pre-doubled so the 64-bit product splits at bit 32: hi = p2>>32 is
// floor(s*A/2^31) and the low dword is 2*(s*A mod 2^31) -- no 64-bit shift needed.
template <int J> __device__ __forceinline__ uint32_t ks_state_byte(uint32_t s)
{
    constexpr uint32_t A2 = 2u * lcg::kBytePow.v[J];
    uint64_t p2 = (uint64_t)s * A2;
    uint32_t X = (uint32_t)(p2 >> 32) + ((uint32_t)p2 >> 1); // non-canonical residue, < 2^32
    return (X + (X >> 31)) & 0xFFu;                          // canonicalise: only bit 31 can be excess
}

template <int J0> __device__ __forceinline__ uint32_t ks_state_dword(uint32_t s)
{
    uint32_t b0 = (J0 == 0) ? (s & 0xFFu) : ks_state_byte<(J0 == 0 ? 1 : J0)>(s);
    uint32_t b1 = ks_state_byte<J0 + 1>(s);
    uint32_t b2 = ks_state_byte<J0 + 2>(s);
    uint32_t b3 = ks_state_byte<J0 + 3>(s);
    return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

// data ^ keystream for one 16-byte word whose first byte has state s.
// keystream = ~state_bytes  =>  data ^ ks = ~(data ^ state_bytes)  (v_xnor_b32).
__device__ __forceinline__ u32x4 cycle_word(u32x4 d, uint32_t s)
{
    d.x = ~(d.x ^ ks_state_dword<0>(s));
    d.y = ~(d.y ^ ks_state_dword<4>(s));
    d.z = ~(d.z ^ ks_state_dword<8>(s));
    d.w = ~(d.w ^ ks_state_dword<12>(s));
    return d;
}

// One byte at state s (head / tail bytes outside the aligned body).
__device__ __forceinline__ uint8_t cycle_byte(uint8_t d, uint32_t s) { return (uint8_t)~(d ^ (uint8_t)s); }

} // namespace

// U = lane-words per thread per loop trip (independent loads in flight per lane).
template <int U>
__global__ __launch_bounds__(lcg::BLOCK) void modgpu_cycle_kernel(CycleArgs a)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;

    // ---- ragged edges: < 16 bytes before / after the aligned body, done bytewise by block 0
    if (blk == 0 && tid < 32) {
        if (tid < a.head_n) {
            uint32_t s = a.base_head;
            for (uint32_t j = 0; j < tid; ++j) s = mulmod_canon(s, lcg::A);
            a.head_ptr[tid] = cycle_byte(a.head_ptr[tid], s);
        } else if (tid >= 16 && tid - 16 < a.tail_n) {
            uint32_t t = tid - 16;
            uint32_t s = a.base_tail;
            for (uint32_t j = 0; j < t; ++j) s = mulmod_canon(s, lcg::A);
            a.tail_ptr[t] = cycle_byte(a.tail_ptr[t], s);
        }
    }

    // ---- aligned body: tiles of 4096 B, U tiles per block per trip, grid-strided
    const uint64_t n_words = a.body_words;
    uint64_t tile = (uint64_t)blk * U;                 // first tile of this block
    const uint64_t tile_step = (uint64_t)gridDim.x * U; // tiles advanced per trip
    uint64_t w = tile * lcg::BLOCK + tid;              // this thread's word in sub-tile 0
    if (w >= n_words) return;

    // jump: base * a^(4096*tile) * a^(16*tid); tile < 65536 (host guarantees gridDim*U <= 65536)
    uint32_t s0 = mulmod_canon(a.base_body, c_tile_hi.v[(tile >> 8) & 255]);
    s0 = mulmod_canon(s0, c_tile_lo.v[tile & 255]);
    s0 = mulmod_canon(s0, c_lane_pow.v[tid]);

    uint32_t s[U];
    s[0] = s0;
#pragma unroll
    for (int u = 1; u < U; ++u) s[u] = mulmod_canon(s[u - 1], lcg::kTileLo.v[1]);

    u32x4 *p = reinterpret_cast<u32x4 *>(a.body) + w;
    const uint64_t word_step = tile_step * lcg::BLOCK;
    const uint64_t last_full = n_words >= (uint64_t)U * lcg::BLOCK ? n_words - (uint64_t)(U - 1) * lcg::BLOCK : 0;

    // full trips: all U sub-tiles in range for this thread  (w + (U-1)*256 < n_words)
    while (w < last_full) {
        u32x4 d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = __builtin_nontemporal_load(p + u * lcg::BLOCK);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = cycle_word(d[u], s[u]);
            s[u] = mulmod_canon(s[u], a.stride_mul);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(d[u], p + u * lcg::BLOCK);
        w += word_step;
        p += word_step;
    }
    // ragged last trip
    if (w < n_words) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (w + (uint64_t)u * lcg::BLOCK < n_words) {
                u32x4 d = p[u * lcg::BLOCK];
                p[u * lcg::BLOCK] = cycle_word(d, s[u]);
            }
        }
    }
}

template __global__ void modgpu_cycle_kernel<1>(CycleArgs);
template __global__ void modgpu_cycle_kernel<2>(CycleArgs);
template __global__ void modgpu_cycle_kernel<4>(CycleArgs);

hipError_t modgpu_launch_cycle(const CycleArgs &a, int unroll, uint32_t grid, hipStream_t stream)
{
    dim3 g(grid), b(lcg::BLOCK);
    switch (unroll) {
    case 1: hipLaunchKernelGGL(modgpu_cycle_kernel<1>, g, b, 0, stream, a); break;
    case 2: hipLaunchKernelGGL(modgpu_cycle_kernel<2>, g, b, 0, stream, a); break;
    default: hipLaunchKernelGGL(modgpu_cycle_kernel<4>, g, b, 0, stream, a); break;
    }
    return hipGetLastError();
}
