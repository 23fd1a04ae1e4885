// cycle_kernel_impl.h -- device code of the cycle kernels (see cycle_kernel.hip for the design notes): the arithmetic,
// and the launch shapes the product ships -- nothing else.  Kept in a header so tools/ can instantiate and time exactly
// this code; every timing ablation, trace hook and rejected variant (copy-only / compute-only loops, the LDS stage,
// other pipeline depths and barrier placements, the plain-C keystream) lives in tools/cycle_kernel_lab.h, which
// includes this file and cannot change it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "lcg.h"
#include "cycle_kernel.h"

namespace {

using u32x4 = uint32_t __attribute__((ext_vector_type(4))); // one dwordx4 lane-word

// ks_word_carry's assembly block works in v[120:127] and s[94:95]: the kernels' own allocation stays below them
#define MODGPU_KEEP_OFF_THE_FIXED_TEMPORARIES __attribute__((amdgpu_num_vgpr(120), amdgpu_num_sgpr(94)))

// Jump tables, baked into the code object (see lcg.h).  kTile* are read with wave-uniform
// indices; kLanePow once per thread at start-up.
__constant__ lcg::Table<256> c_lane_pow = lcg::kLanePow;
__constant__ lcg::Table<256> c_tile_lo = lcg::kTileLo;
__constant__ lcg::Table<256> c_tile_hi = lcg::kTileHi;

// Mersenne fold of the 64-bit product s * (2*y):  p2 = hi * 2^32 + 2 * lo31  with
// hi = floor(s*y / 2^31), lo31 = s*y mod 2^31, so  s*y == hi + lo31 (mod m).
// One more multiply-add does the shift and the add together:
//     lo2 * (2^31 - 1) + p2  =  lo31 * 2^32 - lo2 + hi * 2^32 + lo2  =  (hi + lo31) * 2^32
// i.e. v_mad_u64_u32 {0, X} = p2.lo * m + p2, whose high dword is X = hi + lo31  (< 2^32).
__device__ __forceinline__ uint32_t mul_fold(uint32_t s, uint32_t y2)
{
    uint64_t p2 = (uint64_t)s * y2;
    uint64_t q = (uint64_t)(uint32_t)p2 * lcg::M + p2;
    return (uint32_t)(q >> 32);
}

// Same with the multiplier already doubled (y2 = 2*y < 2^32), e.g. a per-launch constant from the host.
__device__ __forceinline__ uint32_t mulmod_canon2(uint32_t x, uint32_t y2)
{
    uint32_t X = mul_fold(x, y2);
    return min(X, X - lcg::M);
}

// x canonical residue (< 2^31), y any residue < 2^31.  Returns x*y mod m, canonical.
// X = x*y mod m or that + m, never m itself (m is prime, inputs non-zero), so min(X, X - m)
// with unsigned wrap picks the canonical one.
__device__ __forceinline__ uint32_t mulmod_canon(uint32_t x, uint32_t y)
{
    uint32_t X = mul_fold(x, 2u * y);
    return min(X, X - lcg::M);
}

// ---- keystream bytes of one dword --------------------------------------------------------
// ALG 1: canonicalise and pack in one SDWA add per byte: byte J of `w` := low8(X + (X >> 31)).   (small shape)
// ALG 2: as 1, with the "+ (X >> 31)" taken from the fold's carry-out (ks_word_carry, one block per word): no shift.
//        (streaming shapes)
template <int SEL> __device__ __forceinline__ void put_byte(uint32_t &w, uint32_t X)
{
    uint32_t c = X >> 31; // the only possible excess over the canonical residue is m: +1 mod 256
    if constexpr (SEL == 0) // first byte of a dword: the other three are zero-filled, so `w` needs no initial value
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(w) : "v"(X), "v"(c));
    else if constexpr (SEL == 1)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(X), "v"(c));
    else if constexpr (SEL == 2)
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(X), "v"(c));
    else
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(w) : "v"(X), "v"(c));
}

template <int J> __device__ __forceinline__ uint32_t state_x(uint32_t s)
{
    return mul_fold(s, 2u * lcg::kBytePow.v[J]); // non-canonical state of byte J of the word
}

// ALG 2: the canonicalising "+1 if X >= 2^31" comes out of the fold itself, as a carry.
// The product is biased by 2^63 (p2' = s * 2y + 2^63: hi' = hi + 2^31, no overflow since hi < 2^31); the fold
//     q = p2'.lo * m + p2'  =  (hi + lo31 + 2^31) * 2^32  =  (X + 2^31) * 2^32
// then overflows 64 bits exactly when X >= 2^31, and v_mad_u64_u32 delivers that overflow per lane in its scalar
// carry-out (VCC).  q.hi = X ^ 2^31: the low byte is X's.  One v_addc_co_u32_sdwa adds the carry and packs the byte:
// three instructions per byte (mad, mad, addc) instead of four (mad, mad, shift, add) -- 20 issue cycles instead of 24.
// It matters because the kernel is VALU-bound whenever the shader clock sits below ~1.9 GHz, which is where the chip's
// power management puts it for the first ~10 ms after load onset (profiles/r03_first_pass.txt, r03_tune_dvfs.txt).
//
// None of the second half can be said in C++ (no way to ask for the mad's carry-out, none for an SDWA add-with-carry;
// and with a visible bias the compiler ORs it into the high dword with a separate instruction).  clang's inline assembly
// can neither bind an operand to VCC ("{vcc}" is rejected) nor name half of a 64-bit operand, and both are needed (the
// fold multiplies by the product's LOW dword, the addc consumes the fold's HIGH dword, the carry travels in VCC).  So: one
// hand-scheduled block per 16-byte word with FIXED temporaries -- v[120:125] (three product slots, rotating), v[126:127]
// (the fold), s[94:95] (the product mad's unused carry-out, kept away from VCC).  The kernels that use it carry
// amdgpu_num_vgpr(120) / amdgpu_num_sgpr(94), so the register allocator cannot reach these (as plain clobbers -- and even
// as early-clobber physical outputs -- it handed them to inputs of the block; the parity tests caught it); they need
// about 70 VGPRs, and a 1024-thread workgroup may use 128 per lane.  tests/test_capi_cpu.py checks the ISA for both.
// (Cut into per-byte statements with compiler-allocated temporaries, the carry has to detour through an SGPR pair and an
//  s_mov into VCC; that form is correct too but measured slower than ALG 1 in the VALU-bound regime.)  Constraints honoured
// by the schedule:
//   * gfx9 VOP3 reads one scalar operand per instruction: the multiplier is the scalar, bias pair and zero live in VGPRs;
//   * gfx940+: a VALU write of VCC needs 2 wait states before a VALU reads it as carry-in (the compiler inserts s_nop 1
//     there): the product of the byte after next, plus one s_nop 0, sit between each fold and its addc.
//   * gfx940+: a VALU that writes part of a VGPR (SDWA dst_sel) needs 1 wait state before a VALU reads that VGPR (LLVM:
//     hasDstSelForwardingHazard).  Inside the block three instructions sit between two addc's of one dword; the block ENDS
//     with an s_nop 0, because the next instruction is the compiler's (typically the v_xnor on w[3]) and its hazard
//     recognizer cannot see what the block's last instruction was.
// w[0] comes in holding the lane state s (byte 0 of the word is the state itself), w[1..3] are written whole.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm" // "clobber list contains reserved registers: s94, s95" -- reserved by us, for this
__device__ __forceinline__ void ks_word_carry(uint32_t s, uint32_t (&w)[4])
{
    uint32_t zero = 0;
    uint64_t bias = 0x8000000000000000ull;
    asm("" : "+v"(zero), "+v"(bias)); // three VGPRs of constants
    const uint32_t m = lcg::M;
    uint32_t w0 = s, w1, w2, w3;
#define M1(P, Y) "v_mad_u64_u32 v[" P "], s[94:95], %[s], %[" Y "], %[bias]\n\t"
#define M2(PLO, P) "v_mad_u64_u32 v[126:127], vcc, v" PLO ", %[m], v[" P "]\n\t"
#define NOP0 "s_nop 0\n\t"
#define NOP1 "s_nop 1\n\t"
#define AC(W, SEL, UNUSED) "v_addc_co_u32_sdwa %[" W "], vcc, v127, %[zero], vcc dst_sel:BYTE_" SEL " dst_unused:" UNUSED " src0_sel:DWORD src1_sel:DWORD\n\t"
    asm(
        M1("120:121", "y1")
        M1("122:123", "y2")
        M2("120", "120:121") M1("124:125", "y3") NOP0 AC("w0", "1", "UNUSED_PRESERVE")
        M2("122", "122:123") M1("120:121", "y4") NOP0 AC("w0", "2", "UNUSED_PRESERVE")
        M2("124", "124:125") M1("122:123", "y5") NOP0 AC("w0", "3", "UNUSED_PRESERVE")
        M2("120", "120:121") M1("124:125", "y6") NOP0 AC("w1", "0", "UNUSED_PAD")
        M2("122", "122:123") M1("120:121", "y7") NOP0 AC("w1", "1", "UNUSED_PRESERVE")
        M2("124", "124:125") M1("122:123", "y8") NOP0 AC("w1", "2", "UNUSED_PRESERVE")
        M2("120", "120:121") M1("124:125", "y9") NOP0 AC("w1", "3", "UNUSED_PRESERVE")
        M2("122", "122:123") M1("120:121", "y10") NOP0 AC("w2", "0", "UNUSED_PAD")
        M2("124", "124:125") M1("122:123", "y11") NOP0 AC("w2", "1", "UNUSED_PRESERVE")
        M2("120", "120:121") M1("124:125", "y12") NOP0 AC("w2", "2", "UNUSED_PRESERVE")
        M2("122", "122:123") M1("120:121", "y13") NOP0 AC("w2", "3", "UNUSED_PRESERVE")
        M2("124", "124:125") M1("122:123", "y14") NOP0 AC("w3", "0", "UNUSED_PAD")
        M2("120", "120:121") M1("124:125", "y15") NOP0 AC("w3", "1", "UNUSED_PRESERVE")
        M2("122", "122:123") NOP1 AC("w3", "2", "UNUSED_PRESERVE")
        M2("124", "124:125") NOP1 AC("w3", "3", "UNUSED_PRESERVE")
        NOP0 // (the compiler's hazard recognizer does not look inside the block: the wait state a consumer of w3 needs, below)
        : [w0] "+&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3)
        : [s] "v"(s), [bias] "v"(bias), [zero] "v"(zero), [m] "s"(m),
          [y1] "s"(2u * lcg::kBytePow.v[1]), [y2] "s"(2u * lcg::kBytePow.v[2]), [y3] "s"(2u * lcg::kBytePow.v[3]), [y4] "s"(2u * lcg::kBytePow.v[4]),
          [y5] "s"(2u * lcg::kBytePow.v[5]), [y6] "s"(2u * lcg::kBytePow.v[6]), [y7] "s"(2u * lcg::kBytePow.v[7]), [y8] "s"(2u * lcg::kBytePow.v[8]),
          [y9] "s"(2u * lcg::kBytePow.v[9]), [y10] "s"(2u * lcg::kBytePow.v[10]), [y11] "s"(2u * lcg::kBytePow.v[11]), [y12] "s"(2u * lcg::kBytePow.v[12]),
          [y13] "s"(2u * lcg::kBytePow.v[13]), [y14] "s"(2u * lcg::kBytePow.v[14]), [y15] "s"(2u * lcg::kBytePow.v[15])
        : "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s94", "s95", "vcc");
#undef M1
#undef M2
#undef NOP0
#undef NOP1
#undef AC
    w[0] = w0;
    w[1] = w1;
    w[2] = w2;
    w[3] = w3;
}
#pragma clang diagnostic pop

// Low bytes of the canonical states of bytes J0..J0+3 of the word whose first byte has
// canonical state s, packed little-endian (ALG 1).
template <int J0> __device__ __forceinline__ uint32_t ks_state_dword(uint32_t s)
{
    uint32_t w;
    if constexpr (J0 == 0) w = s; // byte 0 of the word is the lane state itself; bytes 1..3 get overwritten
    else put_byte<0>(w, state_x<(J0 == 0 ? 1 : J0)>(s));
    put_byte<1>(w, state_x<J0 + 1>(s));
    put_byte<2>(w, state_x<J0 + 2>(s));
    put_byte<3>(w, state_x<J0 + 3>(s));
    return w;
}

// data ^ keystream for one 16-byte word whose first byte has state s.
// keystream = ~state_bytes  =>  data ^ ks = ~(data ^ state_bytes)  (one v_xnor per dword).
template <int ALG> __device__ __forceinline__ u32x4 cycle_word(u32x4 d, uint32_t s)
{
    static_assert(ALG == 1 || ALG == 2, "the two keystream sequences the product ships");
    if constexpr (ALG == 2) {
        uint32_t w[4];
        ks_word_carry(s, w);
        d.x = ~(d.x ^ w[0]);
        d.y = ~(d.y ^ w[1]);
        d.z = ~(d.z ^ w[2]);
        d.w = ~(d.w ^ w[3]);
        return d;
    }
    d.x = ~(d.x ^ ks_state_dword<0>(s));
    d.y = ~(d.y ^ ks_state_dword<4>(s));
    d.z = ~(d.z ^ ks_state_dword<8>(s));
    d.w = ~(d.w ^ ks_state_dword<12>(s));
    return d;
}

// One byte at state s (head / tail bytes outside the aligned body).
__device__ __forceinline__ uint8_t cycle_byte(uint8_t d, uint32_t s) { return (uint8_t)~(d ^ (uint8_t)s); }

// < 16 bytes before / after the aligned body, done bytewise by 32 lanes of one workgroup
__device__ __forceinline__ void cycle_edges(uint8_t *head_ptr, uint32_t head_n, uint32_t base_head, uint8_t *tail_ptr, uint32_t tail_n,
                                            uint32_t base_tail, uint32_t tid)
{
    if (tid < head_n) {
        uint32_t s = base_head;
        for (uint32_t j = 0; j < tid; ++j) s = mulmod_canon(s, lcg::A);
        head_ptr[tid] = cycle_byte(head_ptr[tid], s);
    } else if (tid >= 16 && tid < 32 && tid - 16 < tail_n) {
        uint32_t t = tid - 16;
        uint32_t s = base_tail;
        for (uint32_t j = 0; j < t; ++j) s = mulmod_canon(s, lcg::A);
        tail_ptr[t] = cycle_byte(tail_ptr[t], s);
    }
}
__device__ __forceinline__ void cycle_edges(const CycleArgs &a, uint32_t tid)
{
    cycle_edges(a.head_ptr, a.head_n, a.base_head, a.tail_ptr, a.tail_n, a.base_tail, tid);
}

// a^(CHUNK * c) by the three bytes of a chunk index c < 2^24 (queue kernel: a workgroup's next chunk is
// whatever the ticket counter hands it, so it jumps from the chunk index instead of striding)
template <uint32_t CHUNK> __constant__ lcg::Table<256> c_chunk_pow0 = lcg::make_pow_table<256>(CHUNK);
template <uint32_t CHUNK> __constant__ lcg::Table<256> c_chunk_pow1 = lcg::make_pow_table<256>((uint64_t)CHUNK << 8);
template <uint32_t CHUNK> __constant__ lcg::Table<256> c_chunk_pow2 = lcg::make_pow_table<256>((uint64_t)CHUNK << 16);

constexpr int AUX_NT = 2;   // buffer cache-policy bit: non-temporal (streaming) -- best for the loads (7.0 vs 6.4 TB/s read-only)
constexpr int AUX_SC1 = 16; // system-coherent / write-through -- best for the stores (6.3 vs 5.9 TB/s write-only)

} // namespace
// ---- the one-shot and the static streaming shape ------------------------------------------------------------------
// U      = lane-words per thread per trip (independent 16-byte loads in flight per lane)
// BLOCK  = threads per workgroup; one workgroup trip covers U*BLOCK*16 contiguous bytes
// ALG    = keystream instruction sequence (see cycle_word)
// STREAM = false: load, compute, store per trip -- the small shape (one word per thread, headers ... 256 MiB).
//          true : software pipeline -- the next trip's loads are issued before this trip's arithmetic, behind a scheduling
//                 barrier that keeps the compiler from hoisting arithmetic above them -- with a workgroup barrier in front
//                 of each trip's loads and another in front of its stores, which keep the 16 waves of a workgroup in step
//                 so its 128 KiB of reads and of writes reach HBM as bursts.
//
// Addressing is buffer_load/store_dwordx4 through a per-trip descriptor built from scalars:
// no per-lane 64-bit pointer arithmetic in the loop, and the hardware range check (num_records =
// bytes left in the body, capped at the chunk) drops the lanes past the end of a ragged last
// chunk -- there is no tail branch.  Loads are `nt`, stores `sc1`: measured best per direction
// (profiles/r01_ubench_copy_policies.txt).
template <int U, int BLOCK, int ALG, bool STREAM>
__global__ __launch_bounds__(BLOCK) MODGPU_KEEP_OFF_THE_FIXED_TEMPORARIES void modgpu_cycle_kernel(CycleArgs a)
{
    static_assert(BLOCK % 256 == 0 && BLOCK <= 1024, "BLOCK is a whole number of 4096-byte tiles");
    constexpr uint32_t CHUNK = (uint32_t)U * BLOCK * lcg::WORD; // bytes per workgroup trip
    constexpr uint32_t SUB = BLOCK * lcg::WORD;                 // bytes per sub-step (one load per lane)
    const uint32_t tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;

    // ---- ragged edges: < 16 bytes before / after the aligned body, done bytewise by block 0
    if (blk == 0 && tid < 32) cycle_edges(a, tid);

    // ---- aligned body.  Chunks sit on ABSOLUTE chunk-aligned addresses (a base that is only 16-byte
    // aligned costs 15 % otherwise: every 1 KiB wave access would straddle 128-byte lines), so the
    // chunk grid starts a.lead bytes before the body; offsets below are relative to that origin.
    // Workgroup b takes chunks b, b + G, b + 2G, ...
    const uint64_t lead = a.lead;
    const uint64_t end = lead + a.body_words * lcg::WORD; // one past the body's last byte
    const uint64_t step = (uint64_t)gridDim.x * CHUNK;
    uint64_t off = (uint64_t)blk * CHUNK;
    if (off >= end) return; // uniform for the workgroup

    // jump to this lane's first word: base * a^(4096*tile) * a^(16*(tid%256)),
    // tile = blk*U*(BLOCK/256) + tid/256 < 65536 (host: grid * U * BLOCK/256 <= 65536);
    // a.base_body already carries a^(-lead), so positions count from the chunk origin
    const uint32_t tile = blk * (U * (BLOCK / 256)) + (tid >> 8);
    uint32_t s[U];
    s[0] = mulmod_canon(a.base_body, c_tile_hi.v[(tile >> 8) & 255]);
    s[0] = mulmod_canon(s[0], c_tile_lo.v[tile & 255]);
    s[0] = mulmod_canon(s[0], c_lane_pow.v[tid & 255]);
#pragma unroll
    for (int u = 1; u < U; ++u) s[u] = mulmod_canon(s[u - 1], lcg::kTileLo.v[BLOCK / 256]);

    uint8_t *const origin = static_cast<uint8_t *>(a.body) - lead; // never dereferenced below the body
    const uint32_t voff = tid * lcg::WORD;

    // The first chunk is cut at the front when the body is not chunk-aligned.  Workgroup 0 peels it
    // off here, outside the hot loop: descriptor based at the body, per-lane offset minus `lead`.
    // Lanes in front of the body get a negative offset, which wraps far past num_records, so the
    // hardware range check drops their loads (zeros) and stores -- the same mechanism that trims
    // the last chunk.  Cold code: not unrolled.
    if (blk == 0 && lead != 0) {
        const uint64_t inside = end < CHUNK ? end - lead : CHUNK - lead;
        auto r = __builtin_amdgcn_make_buffer_rsrc(static_cast<uint8_t *>(a.body), 0, (int)inside, 0x00020000);
        uint32_t su = s[0];
#pragma unroll 1
        for (uint32_t u = 0; u < (uint32_t)U; ++u) {
            const uint32_t o = voff + u * SUB - (uint32_t)lead;
            u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, AUX_NT);
            d = cycle_word<ALG>(d, su);
            __builtin_amdgcn_raw_buffer_store_b128(d, r, o, 0, AUX_SC1);
            su = mulmod_canon(su, lcg::kTileLo.v[BLOCK / 256]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = mulmod_canon2(s[u], a.stride_mul2);
        off += step;
        if (off >= end) return;
    }

    // (opaque copy: keeps the compiler from merging this 32-bit multiplier with the peel path's into
    //  a 64-bit scalar pair, which cost one extra v_mad_u64_u32 per word in the hot loop)
    uint32_t stride2 = a.stride_mul2;
    asm("" : "+s"(stride2));

    // every remaining chunk starts inside the body; only the last can be short
    auto rsrc_at = [&](uint64_t o) {
        uint64_t left = o < end ? end - o : 0;
        return __builtin_amdgcn_make_buffer_rsrc(origin + o, 0, (int)(left < CHUNK ? left : CHUNK), 0x00020000);
    };
    auto load = [&](u32x4(&d)[U], uint64_t o) {
        auto r = rsrc_at(o);
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * SUB, 0, AUX_NT);
    };
    auto process_store = [&](u32x4(&d)[U], uint64_t o) {
        auto r = rsrc_at(o);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = cycle_word<ALG>(d[u], s[u]);
            s[u] = mulmod_canon2(s[u], stride2);
        }
        if constexpr (STREAM) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, AUX_SC1);
    };

    if constexpr (!STREAM) {
        for (; off < end; off += step) {
            u32x4 d[U];
            load(d, off);
            process_store(d, off);
        }
    } else {
        // ping-pong: chunk k+1 is in flight while chunk k is computed and stored.  A chunk past
        // the end has a zero-size descriptor: its loads return 0 and its stores are dropped.
        u32x4 d0[U], d1[U];
        load(d0, off);
        while (true) {
            __builtin_amdgcn_s_barrier();
            load(d1, off + step);
            __builtin_amdgcn_sched_barrier(0);
            process_store(d0, off);
            off += step;
            if (off >= end) break;
            __builtin_amdgcn_s_barrier();
            load(d0, off + step);
            __builtin_amdgcn_sched_barrier(0);
            process_store(d1, off);
            off += step;
            if (off >= end) break;
        }
    }
}

// ---- the streaming kernel with a work queue ---------------------------------------------------------
// Same lane layout, bursts and arithmetic as the static streaming shape above (modgpu_cycle_kernel<U, BLOCK, 2, true>);
// what differs is WHICH chunk a workgroup takes next.  With the static map (b, b+G, b+2G ...)
// every workgroup does the same number of trips, but the CUs are not equally fast: the workgroups of every
// other XCD take ~9.6 us per trip, the rest ~11 us (profiles/r02_trace_static_schedule.txt), so half the
// chip idles at the end of every launch while the other half finishes.  Here a workgroup's first two
// chunks are static (b, b+Gm among the Gm main workgroups: no start-up latency) and every later one is a ticket from a global
// counter, fetched at the start of the trip before the one that loads it, so fast CUs simply take more chunks and all of them
// finish within one trip of each other.
//
// The chunks on offer are those of a TABLE of parts (CycleQueueArgs, cycle_kernel.h): a global chunk index g belongs to the
// part p with start[p] <= g < start[p+1], whose origin, end and base state come from the table in the kernel arguments (read
// where it lies with scalar loads).  One buffer is a table of one part; an archive's parts resident on one GPU share a launch
// and pay its fixed cost once.  A workgroup's consecutive chunks almost always lie in one part, so it keeps two cached views --
// the chunk it is loading, the chunk it is finishing -- and walks the table only when a chunk falls outside its view (scalar
// code, a handful of times per launch).  A view holds the part's base state already multiplied by this lane's share of the
// jump (tile and lane powers), so a trip's first state is one multiply by the chunk's power, whatever the part.
//
// Ticket hand-off inside a workgroup: lane 0 issues the returning atomic at the START of a trip (behind the first barrier,
// in front of that trip's load burst, so waiting for it never waits for those loads), publishes the value through an LDS word
// (two, used alternately) before the SAME trip's second barrier -- the trip's arithmetic, ~2 us, hides the atomic's round trip
// -- and every wave reads it after that barrier: it is the chunk the NEXT trip loads.  (Until round 4 the atomic was issued
// behind the second barrier and published a whole trip later: a workgroup was then committed to one chunk more when the
// tickets ran out -- three instead of two -- and that chunk's worth of imbalance at the end of a launch cost 0.5 % at 4 GiB
// and 0.8 % at 411 MB; profiles/r04_tail.txt, rows "TK".)  a.queue[0] is the ticket counter, a.queue[1] counts workgroups that are done; the last one out
// zeroes both, so the pair is clean for the next launch without a memset, and then writes a.queue_seq to the host-visible
// word a.queue_done: the host hands a pair to a new launch only after its previous user has signed off there.
// One chunk of loads is in flight ahead of the one being computed (ping-pong); nt loads, sc1+nt stores; a workgroup barrier
// in front of each trip's load burst and another in front of its store burst (the second is also what orders the ticket
// hand-off).  Deeper pipelines, other barrier placements and cache policies were measured in tools/ and lost.
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) MODGPU_KEEP_OFF_THE_FIXED_TEMPORARIES void modgpu_cycle_queue_kernel(CycleQueueArgs a)
{
    static_assert(BLOCK % 256 == 0 && BLOCK <= 1024, "BLOCK is a whole number of 4096-byte tiles");
    constexpr int ALG = 2;                   // the three-instruction keystream (ks_word_carry)
    constexpr int SAUX = AUX_SC1 | AUX_NT;   // stores: write-through, streaming
    constexpr int DEPTH = 1;                 // chunks of loads in flight ahead of the one being computed
    constexpr uint32_t CHUNK = (uint32_t)U * BLOCK * lcg::WORD;
    constexpr uint32_t SUB = BLOCK * lcg::WORD;
    constexpr int NB = DEPTH + 1;     // register buffers: one being computed, DEPTH being loaded
    constexpr int PREFIX = DEPTH + 1; // static chunks per workgroup: the ticket fetched in trip j is loaded in trip j + 1 and computed in trip j + PREFIX
    const uint32_t tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const uint32_t G = gridDim.x;
    const uint32_t Gm = a.main_groups != 0 && a.main_groups < G ? a.main_groups : G; // main workgroups; [Gm, G) are helpers (below)
    const uint32_t n_parts = a.n_parts;
    const uint32_t total = a.start[kCycleBatchMax]; // (unused entries of start[] hold the total as well)
    // Two LDS words, used alternately: a trip's ticket is written before that trip's barrier and read after it, and
    // the same word is written again two trips later -- i.e. behind the NEXT trip's barrier, which no wave can reach
    // before it has done this trip's read.  (With a single word, correctness would lean on the other barrier, the
    // one in front of the loads, which is a tuning choice: without it a wave held up between this barrier and its
    // read can be overtaken by lane 0's next write -- tools/tune_cycle's INVALID row shows what that looks like.)
    __shared__ uint32_t q_next[2];
    uint32_t trip = 0;
    const uint32_t voff = tid * lcg::WORD;
    // a^(4096*(tid/256)) * a^(16*(tid%256)): this lane's word 0 relative to the start of any chunk
    const uint32_t lane_mul = mulmod_canon(c_tile_lo.v[tid >> 8], c_lane_pow.v[tid & 255]);

    // Ragged edges (< 16 bytes before / after a part's aligned body) and the part's first chunk when the body is not
    // chunk-aligned: workgroup p does them for part p, before the stream starts (cold code).  Chunks sit on ABSOLUTE
    // chunk-aligned addresses, so that first chunk is cut at the front: descriptor based at the body, per-lane offset minus
    // `lead`; lanes in front of the body get a negative offset, which wraps far past num_records, so the hardware range check
    // drops their loads and stores.  It is not part of the chunk index space.
    for (uint32_t p = blk; p < n_parts; p += G) {
        const CycleQueuePart &P = a.part[p];
        const uint64_t body_bytes = P.end - P.lead;
        if (tid < 32) cycle_edges(P.body - P.head_n, P.head_n, P.base_head, P.body + body_bytes, P.tail_n, P.base_tail, tid);
        if (P.lead != 0 && body_bytes != 0) {
            const uint64_t inside = P.end < CHUNK ? body_bytes : CHUNK - P.lead;
            auto r = __builtin_amdgcn_make_buffer_rsrc(P.body, 0, (int)inside, 0x00020000);
            uint32_t su = mulmod_canon(P.base_body, lane_mul);
#pragma unroll 1
            for (uint32_t u = 0; u < (uint32_t)U; ++u) {
                const uint32_t o = voff + u * SUB - P.lead;
                u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, AUX_NT);
                d = cycle_word<ALG>(d, su);
                __builtin_amdgcn_raw_buffer_store_b128(d, r, o, 0, SAUX);
                su = mulmod_canon(su, lcg::kTileLo.v[BLOCK / 256]);
            }
        }
    }

    // the part a global chunk index lies in, as far as the loop needs it
    struct View {
        uint8_t *origin;    // body - lead
        uint64_t end;
        uint32_t lo, hi;    // global indices [lo, hi) map to the part's chunks first + (g - lo)
        uint32_t first;     // 1 if the part's chunk 0 is the cut one (done above)
        uint32_t lane_base; // per lane: state of this lane's word 0 in the part's chunk 0
    };
    auto locate = [&](uint32_t g, View &v) {
        if (g - v.lo < v.hi - v.lo) return; // lo <= g < hi
        uint32_t p = 0;
#pragma unroll 1
        for (uint32_t i = 1; i < n_parts; ++i) p += g >= a.start[i] ? 1u : 0u; // (empty parts share their start with the next one: skipped)
        const CycleQueuePart &P = a.part[p];
        v.origin = P.body - P.lead;
        v.end = P.end;
        v.first = P.lead != 0 ? 1u : 0u;
        v.lo = a.start[p];
        v.hi = a.start[p + 1];
        v.lane_base = mulmod_canon(P.base_body, lane_mul);
    };
    auto rsrc_at = [&](uint32_t g, const View &v) {
        const uint64_t o = (uint64_t)(v.first + (g - v.lo)) * CHUNK;
        const uint64_t left = g < v.hi && o < v.end ? v.end - o : 0; // past the last part: zero-size descriptor, loads give 0, stores drop
        return __builtin_amdgcn_make_buffer_rsrc(v.origin + o, 0, (int)(left < CHUNK ? left : CHUNK), 0x00020000);
    };
    // states of this lane's U words in chunk g: the part's chunk c multiplies lane_base by a^(CHUNK*c), c < 2^24 (host)
    auto states = [&](uint32_t g, const View &v, uint32_t(&s)[U]) {
        const uint32_t c = v.first + (g - v.lo);
        uint32_t p = mulmod_canon(c_chunk_pow0<CHUNK>.v[c & 255], c_chunk_pow1<CHUNK>.v[(c >> 8) & 255]);
        p = mulmod_canon(p, c_chunk_pow2<CHUNK>.v[(c >> 16) & 255]);
        s[0] = mulmod_canon(v.lane_base, p);
#pragma unroll
        for (int u = 1; u < U; ++u) s[u] = mulmod_canon(s[u - 1], lcg::kTileLo.v[BLOCK / 256]);
    };
    View vl{nullptr, 0, 0, 0, 0, 1}, vs{nullptr, 0, 0, 0, 0, 1}; // load side, store side
    auto load = [&](u32x4(&d)[U], uint32_t g) {
        locate(g, vl);
        auto r = rsrc_at(g, vl);
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * SUB, 0, AUX_NT);
    };
    // Lane 0's ticket traffic.  The returning atomic is a plain compiler-visible atomic, so the compiler counts it
    // in its own s_waitcnt vmcnt(N) bookkeeping and waits for the value only where it is published, behind the
    // trip's arithmetic (with the next chunk's loads, issued after it, still in flight).  This needs the TU built with  -mllvm -amdgpu-atomic-optimizer-strategy=None : the default
    // "atomic optimizer" rewrites it into a wave-aggregated atomic followed at once by s_waitcnt vmcnt(0),
    // i.e. the wave would sit out the atomic's round trip and every load it has in flight, each trip.
    // The LDS word is accessed with ds_write / ds_read in assembly: a volatile C++ access to a __shared__
    // variable becomes a FLAT access, which waits on vmcnt as well as lgkmcnt.
    uint32_t pending = 0; // lane 0: the ticket in flight
    const uint32_t q_next_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&q_next[0];
    const uint32_t one = 1u;
    // one trip: chunk g's words are in d; compute, publish the ticket fetched at the start of this trip, barrier, store burst
    auto process_store = [&](u32x4(&d)[U], uint32_t g) {
        locate(g, vs);
        auto r = rsrc_at(g, vs);
        uint32_t s[U];
        states(g, vs, s);
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = cycle_word<ALG>(d[u], s[u]);
        if (tid == 0) // (the LDS write has landed before the barrier releases the readers)
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(q_next_lds + 4u * (trip & 1u)), "v"(pending) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, SAUX);
        ++trip;
    };
    auto take_published = [&]() { // every lane, after the trip's barrier (trip already counted: the word is (trip-1)&1)
        uint32_t t;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(q_next_lds + 4u * ((trip - 1u) & 1u)) : "memory");
        return (uint32_t)PREFIX * Gm + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };

    // A workgroup's chunk sequence: positions 0 .. PREFIX-1 are static (b, b+Gm), position j + PREFIX is the
    // ticket fetched in trip j.  cq[] holds positions k .. k+DEPTH at the start of trip k: cq[0] is computed,
    // cq[DEPTH] is loaded now, the ones between are already in flight.
    uint32_t cq[NB];
    static_assert(PREFIX == NB, "the static positions are exactly the ones cq[] starts with");
    bool active = true;
    if (blk < Gm) {
#pragma unroll
        for (int i = 0; i < NB; ++i) cq[i] = blk + (uint32_t)i * Gm;
    } else {
        // A HELPER workgroup.  With the chip's clock where it normally is (2.1-2.2 GHz) the 25-per-32-CU main workgroups
        // saturate HBM and more streams only hurt (-1.8 % at one per CU).  For the first ~10 ms after load onset,
        // though, power management holds the shader clock at 1.2-1.7 GHz, and there the main workgroups run out of
        // ARITHMETIC (profiles/r03_first_pass.txt): the idle CUs' SIMDs are then worth more than the tidy memory
        // pattern (flat 6.85 TB/s with a workgroup on every CU against a dip to 6.2-6.4).  So the idle CUs get a
        // workgroup each that looks at the clock ONCE, when it starts -- shader-clock ticks (s_memtime) per 2 us of the
        // constant 100 MHz counter (s_memrealtime) -- and either joins, taking its first PREFIX chunks and all later
        // ones from the ticket counter, or leaves at once.  (Helpers that stay and keep watching the clock were tried:
        // correct, but with 56 workgroups standing by the main ones ran 15 % slower at full clock --
        // profiles/r03_tune_dvfs.txt keeps that row.)
        uint32_t t = 0xFFFFFFFFu;
        if (tid == 0) {
            const uint64_t t0 = wall_clock64(), c0 = clock64();
            uint64_t t1;
            do {
                __builtin_amdgcn_s_sleep(4);
                t1 = wall_clock64();
            } while (t1 - t0 < 200);
            const uint64_t mhz = ((clock64() - c0) * 100) / (t1 - t0);
            if (mhz < a.helper_below_mhz) t = __hip_atomic_fetch_add(a.queue, (uint32_t)PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q_next[0] = t;
        }
        __syncthreads();
        t = q_next[0];
        __syncthreads(); // (the loop below writes q_next[0] again, in its first trip)
        active = t != 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NB; ++i) cq[i] = (uint32_t)PREFIX * Gm + t + (uint32_t)i;
    }
    if (active && cq[0] < total) {
        u32x4 d[NB][U];
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) load(d[i], cq[i]);
        bool finished = false;
        while (!finished) {
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                __builtin_amdgcn_s_barrier();
                // the ticket for the NEXT trip's load burst: fetched now, published behind this trip's arithmetic
                if (tid == 0) pending = __hip_atomic_fetch_add(a.queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                load(d[(p + DEPTH) % NB], cq[DEPTH]);
                __builtin_amdgcn_sched_barrier(0);
                process_store(d[p], cq[0]);
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) cq[i] = cq[i + 1];
                cq[DEPTH] = take_published();
                if (cq[0] >= total) {
                    finished = true;
                    break;
                }
            }
        }
    }
    // leave: this workgroup's ticket atomics have all returned; the last workgroup out resets the pair and then
    // tells the host (a word in host-coherent memory) that the pair may be handed to another launch
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(a.queue + 1, 1u) == G - 1) {
            __hip_atomic_store(a.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.queue_done) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // both zeroes have been performed device-wide
                __hip_atomic_store(a.queue_done, a.queue_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
