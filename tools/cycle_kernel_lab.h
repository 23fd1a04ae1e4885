// tools/cycle_kernel_lab.h -- the cycle kernels WITH their tuning knobs: timing ablations (copy-only / compute-only loops),
// per-workgroup trace stamps, the LDS write-combine stage, other pipeline depths, barrier placements, cache policies and
// keystream sequences, and the tail experiments of round 4.  tools/ only: the product header
// (modulate_amd/csrc/cycle_kernel_impl.h) holds the shipped shapes and nothing else, this file includes it for the
// arithmetic (mul_fold, cycle_word, cycle_edges, the jump tables) and cannot change it.  Some settings here are
// KNOWINGLY WRONG (B2 = 0 races the ticket hand-off: timing only); tools/tune_cycle marks such rows INVALID.
// With every knob at its default a lab kernel is the product kernel's loop: tools/tune_cycle times both side by side.
#pragma once
#include <type_traits>
#include "cycle_kernel_impl.h"

constexpr uint32_t kTraceSlots = 64; // per workgroup: start, up to 62 trip ends, XCC id
namespace {
enum : int { MODE_FULL = 0, MODE_COPY = 1, MODE_COMPUTE = 2 };

// ALG 0: plain C; the compiler packs the four low bytes with shifts / v_perm.  ALG 1, 2: the product's (cycle_word).
template <int J0> __device__ __forceinline__ uint32_t lab_ks_state_dword_c(uint32_t s)
{
    auto lowbyte = [](uint32_t X) { return (X + (X >> 31)) & 0xFFu; };
    uint32_t b0 = (J0 == 0) ? (s & 0xFFu) : lowbyte(state_x<(J0 == 0 ? 1 : J0)>(s));
    return b0 | (lowbyte(state_x<J0 + 1>(s)) << 8) | (lowbyte(state_x<J0 + 2>(s)) << 16) | (lowbyte(state_x<J0 + 3>(s)) << 24);
}
template <int ALG> __device__ __forceinline__ u32x4 lab_cycle_word(u32x4 d, uint32_t s)
{
    if constexpr (ALG == 0) {
        d.x = ~(d.x ^ lab_ks_state_dword_c<0>(s));
        d.y = ~(d.y ^ lab_ks_state_dword_c<4>(s));
        d.z = ~(d.z ^ lab_ks_state_dword_c<8>(s));
        d.w = ~(d.w ^ lab_ks_state_dword_c<12>(s));
        return d;
    } else return cycle_word<ALG>(d, s);
}
} // namespace

// One buffer as tools/ describe it: the product's plan plus what only the lab kernels take.
struct LabArgs : CycleArgs {
    uint32_t *queue = nullptr, *queue_done = nullptr; // {ticket counter, workgroups done}
    uint32_t queue_seq = 0, main_groups = 0, helper_below_mhz = 0;
    uint32_t tail_chunks = 0; // TSPLIT != 0: this many chunks at the end of the index space are handed out as 2^TSPLIT pieces each
    uint32_t standby_ticks = 0; // see LabQueueArgs
    uint64_t *trace = nullptr; // TRACE = 1: per-workgroup timestamps (wall_clock64, 100 MHz), kTraceSlots words each: [0] start, [1+k] end of trip k, [last] XCC id
};
struct LabQueueArgs {
    CycleQueueArgs q;
    uint64_t *trace;
    uint32_t tail_chunks;
    uint32_t standby_ticks; // round 5, helpers: 0 = look at the clock once and join or leave (the product); else keep looking, every ~50 us, for this
                            // many ticks of the 100 MHz counter, sleeping in between WITHOUT touching memory, and join the moment the clock is low
};
// the product's table of one part for a buffer planned as CycleArgs (chunk = the instantiation's bytes per workgroup trip)
inline CycleQueueArgs lab_queue_table_of(const LabArgs &a, uint32_t chunk)
{
    CycleQueueArgs q{};
    q.queue = a.queue;
    q.queue_done = a.queue_done;
    q.queue_seq = a.queue_seq;
    q.main_groups = a.main_groups;
    q.helper_below_mhz = a.helper_below_mhz;
    q.n_parts = 1;
    CycleQueuePart &P = q.part[0];
    P.body = static_cast<uint8_t *>(a.body);
    P.lead = a.lead;
    P.end = (uint64_t)a.lead + a.body_words * 16;
    P.base_body = a.base_body;
    P.base_head = a.base_head;
    P.base_tail = a.base_tail;
    P.head_n = a.head_n;
    P.tail_n = a.tail_n;
    const uint64_t n_chunks = (P.end + chunk - 1) / chunk, first = a.lead != 0 ? 1 : 0;
    const uint32_t total = (uint32_t)(n_chunks > first ? n_chunks - first : 0);
    for (int k = 1; k <= kCycleBatchMax; ++k) q.start[k] = total;
    return q;
}
inline LabQueueArgs lab_queue_args_of(const LabArgs &a, uint32_t chunk) { return {lab_queue_table_of(a, chunk), a.trace, a.tail_chunks, a.standby_ticks}; }

// ---- the static-map kernel with every knob ------------------------------------------------------------------
// PIPE  = 0: load, compute, store per trip.  1: software pipeline, the next trip's loads are issued
//         before this trip's arithmetic.  2: same, with a scheduling barrier that keeps the compiler
//         from hoisting arithmetic above those loads (the product's streaming shape).  3: as 2, and each word is stored
//         as soon as it is finished instead of all U at the end of the trip
// SYNC  = bit 0: workgroup barrier before each trip's loads, bit 1: before its stores
// MODE  = MODE_FULL: the cipher; MODE_COPY / MODE_COMPUTE: timing ablations
// TRACE = 1: lane 0 of every workgroup records when it started and when each trip's stores had been issued
// LDSW  = 1 (the north_star's "LDS as a write-combine stage", measured, not shipped): each trip's finished words go
//         registers -> LDS -> registers before the store burst
template <int U, int BLOCK, int ALG, int PIPE, int MODE, int SAUX = AUX_SC1, int SYNC = 0, int TRACE = 0, int LDSW = 0>
__global__ __launch_bounds__(BLOCK) MODGPU_KEEP_OFF_THE_FIXED_TEMPORARIES void lab_cycle_kernel(LabArgs a)
{
    static_assert(BLOCK % 256 == 0 && BLOCK <= 1024, "BLOCK is a whole number of 4096-byte tiles");
    constexpr uint32_t CHUNK = (uint32_t)U * BLOCK * lcg::WORD; // bytes per workgroup trip
    constexpr uint32_t SUB = BLOCK * lcg::WORD;                 // bytes per sub-step (one load per lane)
    const uint32_t tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    [[maybe_unused]] uint32_t trip = 0;
    [[maybe_unused]] auto stamp = [&](uint32_t slot) {
        if constexpr (TRACE != 0) {
            if (tid == 0 && slot < kTraceSlots - 1) a.trace[blk * kTraceSlots + slot] = wall_clock64();
        }
    };
    if constexpr (TRACE != 0) {
        if (tid == 0) a.trace[blk * kTraceSlots + kTraceSlots - 1] = __builtin_amdgcn_s_getreg((20 | (0 << 6) | (3 << 11))); // HW_REG_XCC_ID[3:0]
        stamp(0);
    }

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
            if constexpr (MODE == MODE_COPY) d = ~d;
            else d = lab_cycle_word<ALG>(d, su);
            if constexpr (MODE != MODE_COMPUTE) __builtin_amdgcn_raw_buffer_store_b128(d, r, o, 0, SAUX);
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
        for (int u = 0; u < U; ++u) {
            if constexpr (MODE == MODE_COMPUTE) d[u] = u32x4{tid, blk, (uint32_t)o, (uint32_t)u};
            else d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * SUB, 0, AUX_NT);
        }
    };
    auto process_store = [&](u32x4(&d)[U], uint64_t o) {
        auto r = rsrc_at(o);
        if constexpr (PIPE == 3 && MODE == MODE_FULL) { // store each word as soon as it is done
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = lab_cycle_word<ALG>(d[u], s[u]);
                __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, SAUX);
                s[u] = mulmod_canon2(s[u], stride2);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (MODE == MODE_COPY) d[u] = ~d[u];
            else {
                d[u] = lab_cycle_word<ALG>(d[u], s[u]);
                s[u] = mulmod_canon2(s[u], stride2);
            }
        }
        if constexpr (LDSW != 0) {
            // write-combine stage: sub-step u's 16 KiB (BLOCK x 16 B) sits contiguously in LDS exactly as it
            // will sit in HBM, then every lane reads its own word back.  Nothing is re-ordered -- the
            // register layout is already the store layout -- so this measures the stage's price.
            __shared__ u32x4 stage[U * BLOCK];
#pragma unroll
            for (int u = 0; u < U; ++u) stage[u * BLOCK + tid] = d[u];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = stage[u * BLOCK + tid];
        }
        if constexpr ((SYNC & 2) != 0 && PIPE != 0) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (MODE == MODE_COMPUTE) {
                if ((d[u].x ^ d[u].y ^ d[u].z ^ d[u].w) == 0x9E3779B9u && s[u] == 1u)
                    __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, SAUX);
            } else __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, SAUX);
        }
        stamp(++trip);
    };

    if constexpr (PIPE == 0) {
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
            if constexpr (SYNC & 1) __builtin_amdgcn_s_barrier();
            load(d1, off + step);
            if constexpr (PIPE >= 2) __builtin_amdgcn_sched_barrier(0);
            process_store(d0, off);
            off += step;
            if (off >= end) break;
            if constexpr (SYNC & 1) __builtin_amdgcn_s_barrier();
            load(d0, off + step);
            if constexpr (PIPE >= 2) __builtin_amdgcn_sched_barrier(0);
            process_store(d1, off);
            off += step;
            if (off >= end) break;
        }
    }
}

// ---- the work-queue kernel with every knob --------------------------------------------------------------------
// (the schedule is described at modgpu_cycle_queue_kernel in the product header)
// DEPTH = chunks of loads a workgroup keeps in flight ahead of the one it computes (1 = ping-pong, the product)
// MODE  = MODE_COPY: the same loop without the keystream -- the memory system's ceiling for this access pattern
// LAUX  = cache-policy bits of the loads (AUX_NT in the product); B1 = 0 drops the barrier in front of each trip's load burst
// B2    = 0 (TIMING ONLY -- results are wrong): drops the second barrier too, so the ticket hand-off races; answers what
//         a barrier-free workgroup would gain.  B2 = 2: the barrier sits behind the store burst instead of in front of it
// TSPLIT (round 4, ONE part only): the last la.tail_chunks chunks of the index space are handed out as 2^TSPLIT pieces each
//         (1: halves, 2: quarters) -- a finer grain where the launch runs out of work, from the same single ticket counter.
//         A piece is loaded with the chunk's U loads through a descriptor of the piece's size (the hardware drops the words
//         beyond it) and only its U >> TSPLIT words are computed.
// TLOOP = 1 (round 4, with TSPLIT >= 1, ONE part): the pieces are NOT handled inside the trip loop.  The trip loop runs over the whole
//         chunks only, exactly as the product's (its index space ends at n_full); a workgroup whose next position lies beyond that
//         falls into a SECOND, cold loop of the same shape over pieces -- same ticket counter, same mailbox protocol, the positions
//         it already holds carried over -- so the finer grain at the tail costs the hot loop nothing but the branch at its exit.
// TK    = 1 (round 4; the PRODUCT's since then): the ticket is fetched at the START of a trip (in front of the load burst) and
//         published in the same trip, so a workgroup is committed to one chunk fewer when the tickets run out (PREFIX = DEPTH + 1).
//         TK = 0: round 3's timing (fetched behind the second barrier, published a trip later)
// LSP   (round 5, one counter-guided experiment: profiles/r04_memside_counters.json has reads in flight at a third of what the
//         workgroups could keep outstanding -- each has loads in flight for only ~1/3 of its 3.7 us trip).  Spreads the next chunk's
//         U loads over the trip instead of issuing them as one burst behind the first barrier:
//         1 = two half-bursts: U/2 loads at the trip's start, U/2 behind the keystream of word U/2 - 1 (both barriers kept);
//         2 = the first U/2 words of chunk k+2 are issued at the END of trip k, behind its store burst and the ticket read (the
//             buffer they land in has just been stored), the other U/2 behind trip k+1's first barrier as before.
//         DEPTH 1, TK 1, no TSPLIT.
// HSB   (round 5) = 1: helper workgroups STAND BY instead of deciding once: lane 0 sleeps (s_sleep and the real-time counter only, no memory
//         traffic), looks at the shader clock again every ~50 us for la.standby_ticks ticks of the 100 MHz counter, and joins the moment it is low.
template <int U, int BLOCK, int ALG, int SAUX = AUX_SC1, int TRACE = 0, int DEPTH = 1, int MODE = MODE_FULL, int LAUX = AUX_NT, int B1 = 1, int B2 = 1,
          int TSPLIT = 0, int TK = 1, int TLOOP = 0, int LSP = 0, int HSB = 0>
__global__ __launch_bounds__(BLOCK) MODGPU_KEEP_OFF_THE_FIXED_TEMPORARIES void lab_cycle_queue_kernel(LabQueueArgs la)
{
    static_assert(LSP == 0 || (DEPTH == 1 && TK == 1 && TSPLIT == 0 && TLOOP == 0 && U % 2 == 0), "LSP is an experiment on the product's loop shape");
    const CycleQueueArgs &a = la.q;
    static_assert(BLOCK % 256 == 0 && BLOCK <= 1024, "BLOCK is a whole number of 4096-byte tiles");
    static_assert(DEPTH >= 1 && DEPTH <= 3, "1..3 chunks of loads in flight");
    constexpr uint32_t CHUNK = (uint32_t)U * BLOCK * lcg::WORD;
    constexpr uint32_t SUB = BLOCK * lcg::WORD;
    constexpr int NB = DEPTH + 1;     // register buffers: one being computed, DEPTH being loaded
    constexpr int PREFIX = TK != 0 ? DEPTH + 1 : DEPTH + 2; // static chunks per workgroup: a ticket fetched in trip j feeds trip j + PREFIX
    static_assert(TSPLIT >= 0 && (U >> TSPLIT) >= 1, "a piece is at least one word per thread");
    const uint32_t tid = threadIdx.x;
    const uint32_t blk = blockIdx.x;
    const uint32_t G = gridDim.x;
    const uint32_t Gm = a.main_groups != 0 && a.main_groups < G ? a.main_groups : G; // main workgroups; [Gm, G) are helpers (below)
    const uint32_t n_parts = a.n_parts;
    const uint32_t total_chunks = a.start[kCycleBatchMax]; // (unused entries of start[] hold the total as well)
    // index space: [0, n_full) whole chunks, then the tail chunks' pieces
    const uint32_t n_full = TSPLIT != 0 && la.tail_chunks < total_chunks ? total_chunks - la.tail_chunks : total_chunks;
    const uint32_t total = n_full + ((total_chunks - n_full) << TSPLIT);
    constexpr bool INLOOP = TSPLIT != 0 && TLOOP == 0; // pieces handled by the trip loop itself (the first form measured)
    static_assert(TLOOP == 0 || (TSPLIT >= 1 && DEPTH == 1 && MODE == MODE_FULL && TK == 0), "the cold tail loop exists for the product's loop shape");
    const uint32_t limit_main = TLOOP != 0 ? n_full : total; // where the trip loop's index space ends
    // Two LDS words, used alternately: a trip's ticket is written before that trip's barrier and read after it, and
    // the same word is written again two trips later -- i.e. behind the NEXT trip's barrier, which no wave can reach
    // before it has done this trip's read.  (With a single word, correctness would lean on the other barrier, the
    // one in front of the loads, which is a tuning choice: without it a wave held up between this barrier and its
    // read can be overtaken by lane 0's next write -- tools/tune_cycle's INVALID row shows what that looks like.)
    __shared__ uint32_t q_next[2];
    uint32_t trip = 0;
    [[maybe_unused]] auto stamp = [&](uint32_t slot) {
        if constexpr (TRACE != 0) {
            if (tid == 0 && slot < kTraceSlots - 1) la.trace[blk * kTraceSlots + slot] = wall_clock64();
        }
    };
    if constexpr (TRACE != 0) {
        if (tid == 0) la.trace[blk * kTraceSlots + kTraceSlots - 1] = __builtin_amdgcn_s_getreg((20 | (0 << 6) | (3 << 11)));
        stamp(0);
    }
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
                d = lab_cycle_word<ALG>(d, su);
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
        v.hi = INLOOP ? total : TLOOP != 0 ? n_full : a.start[p + 1];
        v.lane_base = mulmod_canon(P.base_body, lane_mul);
    };
    // global index -> the part's chunk, and which piece of it (TSPLIT: pieces of CHUNK >> TSPLIT bytes beyond n_full)
    auto piece_of = [&](uint32_t g, const View &v, uint32_t *c, uint32_t *piece) {
        const uint32_t k = g - v.lo;
        if (!INLOOP || k < n_full) {
            *c = v.first + k;
            *piece = 0;
            return false;
        }
        const uint32_t h = k - n_full;
        *c = v.first + n_full + (h >> TSPLIT);
        *piece = h & ((1u << TSPLIT) - 1u);
        return true;
    };
    auto rsrc_at = [&](uint32_t g, const View &v) {
        uint32_t c, piece;
        const bool split = piece_of(g, v, &c, &piece);
        const uint64_t o = (uint64_t)c * CHUNK + (uint64_t)piece * (CHUNK >> TSPLIT);
        const uint64_t cap = split ? (CHUNK >> TSPLIT) : CHUNK;
        const uint64_t left = g < v.hi && o < v.end ? v.end - o : 0; // past the last part: zero-size descriptor, loads give 0, stores drop
        return __builtin_amdgcn_make_buffer_rsrc(v.origin + o, 0, (int)(left < cap ? left : cap), 0x00020000);
    };
    // states of this lane's U words in chunk g: the part's chunk c multiplies lane_base by a^(CHUNK*c), c < 2^24 (host)
    auto states = [&](uint32_t g, const View &v, uint32_t(&s)[U]) {
        uint32_t c, piece;
        const bool split = piece_of(g, v, &c, &piece);
        uint32_t p = mulmod_canon(c_chunk_pow0<CHUNK>.v[c & 255], c_chunk_pow1<CHUNK>.v[(c >> 8) & 255]);
        p = mulmod_canon(p, c_chunk_pow2<CHUNK>.v[(c >> 16) & 255]);
        if (split) p = mulmod_canon(p, c_tile_lo.v[piece * ((CHUNK >> TSPLIT) / 4096u)]); // a^(piece * piece bytes)
        s[0] = mulmod_canon(v.lane_base, p);
#pragma unroll
        for (int u = 1; u < U; ++u) s[u] = mulmod_canon(s[u - 1], lcg::kTileLo.v[BLOCK / 256]);
    };
    View vl{nullptr, 0, 0, 0, 0, 1}, vs{nullptr, 0, 0, 0, 0, 1}; // load side, store side
    auto load = [&](u32x4(&d)[U], uint32_t g) {
        locate(g, vl);
        auto r = rsrc_at(g, vl);
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * SUB, 0, LAUX);
    };
    // LSP: words [U0, U1) of chunk g only
    [[maybe_unused]] auto load_some = [&](u32x4(&d)[U], uint32_t g, auto u0, auto u1) {
        locate(g, vl);
        auto r = rsrc_at(g, vl);
#pragma unroll
        for (int u = decltype(u0)::value; u < decltype(u1)::value; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * SUB, 0, LAUX);
    };
    // Lane 0's ticket traffic.  The returning atomic is a plain compiler-visible atomic, so the compiler counts it
    // in its own s_waitcnt vmcnt(N) bookkeeping and waits for the value only where it is published, a trip
    // later.  This needs the TU built with  -mllvm -amdgpu-atomic-optimizer-strategy=None : the default
    // "atomic optimizer" rewrites it into a wave-aggregated atomic followed at once by s_waitcnt vmcnt(0),
    // i.e. the wave would sit out the atomic's round trip and every load it has in flight, each trip.
    // The LDS word is accessed with ds_write / ds_read in assembly: a volatile C++ access to a __shared__
    // variable becomes a FLAT access, which waits on vmcnt as well as lgkmcnt.
    uint32_t pending = 0; // lane 0: the ticket in flight
    const uint32_t q_next_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&q_next[0];
    const uint32_t one = 1u;
    // one trip: chunk g's words are in d; compute, publish last trip's ticket, barrier, fetch a ticket, store burst
    auto process_store = [&](u32x4(&d)[U], uint32_t g, bool publish, auto &&mid) { // mid(): issued between words U/2 - 1 and U/2 (LSP 1)
        locate(g, vs);
        auto r = rsrc_at(g, vs);
        if constexpr (MODE == MODE_COPY) {
#pragma unroll
            for (int u = 0; u < U / 2; ++u) d[u] = ~d[u];
            mid();
#pragma unroll
            for (int u = U / 2; u < U; ++u) d[u] = ~d[u];
        } else {
            uint32_t s[U];
            states(g, vs, s);
            if (INLOOP && g - vs.lo >= n_full) { // a piece: its words only (wave-uniform branch)
#pragma unroll
                for (int u = 0; u < (U >> TSPLIT); ++u) d[u] = lab_cycle_word<ALG>(d[u], s[u]);
            } else if constexpr (LSP == 1) {
#pragma unroll
                for (int u = 0; u < U / 2; ++u) d[u] = lab_cycle_word<ALG>(d[u], s[u]);
                __builtin_amdgcn_sched_barrier(0);
                mid();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = U / 2; u < U; ++u) d[u] = lab_cycle_word<ALG>(d[u], s[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) d[u] = lab_cycle_word<ALG>(d[u], s[u]);
            }
        }
        if (publish && tid == 0) // (the LDS write has landed before the barrier releases the readers)
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(q_next_lds + 4u * (trip & 1u)), "v"(pending) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (B2 == 1) __builtin_amdgcn_s_barrier();
        if constexpr (TK == 0) {
            if (tid == 0) pending = __hip_atomic_fetch_add(a.queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, SAUX);
        if constexpr (B2 == 2) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        ++trip;
        stamp(trip);
    };
    auto take_published = [&]() { // every lane, after the trip's barrier (trip already counted: the word is (trip-1)&1)
        uint32_t t;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(q_next_lds + 4u * ((trip - 1u) & 1u)) : "memory");
        return (uint32_t)PREFIX * Gm + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };

    // A workgroup's chunk sequence: positions 0 .. PREFIX-1 are static (b, b+Gm, ...), position j + PREFIX is the
    // ticket fetched in trip j.  cq[] holds positions k .. k+DEPTH at the start of trip k: cq[0] is computed,
    // cq[DEPTH] is loaded now, the ones between are already in flight.
    uint32_t cq[NB];
    uint32_t last_static; // position PREFIX-1, enters cq after trip 0
    bool active = true;
    if (blk < Gm) {
#pragma unroll
        for (int i = 0; i < NB; ++i) cq[i] = blk + (uint32_t)i * Gm;
        last_static = blk + (uint32_t)NB * Gm;
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
            if constexpr (HSB == 0) { // the product's: look once
                const uint64_t t0 = wall_clock64(), c0 = clock64();
                uint64_t t1;
                do {
                    __builtin_amdgcn_s_sleep(4);
                    t1 = wall_clock64();
                } while (t1 - t0 < 200);
                const uint64_t mhz = ((clock64() - c0) * 100) / (t1 - t0);
                if (mhz < a.helper_below_mhz) t = __hip_atomic_fetch_add(a.queue, (uint32_t)PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if constexpr (HSB == 2) { // diagnostic: sleep la.standby_ticks, then join whatever the clock (what does a LATE joiner do to a launch?)
                const uint64_t born = wall_clock64();
                while (wall_clock64() - born < la.standby_ticks) __builtin_amdgcn_s_sleep(127);
                t = __hip_atomic_fetch_add(a.queue, (uint32_t)PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                const uint64_t born = wall_clock64();
                for (;;) {
                    const uint64_t t0 = wall_clock64(), c0 = clock64();
                    uint64_t t1;
                    do {
                        __builtin_amdgcn_s_sleep(4);
                        t1 = wall_clock64();
                    } while (t1 - t0 < 200);
                    const uint64_t mhz = ((clock64() - c0) * 100) / (t1 - t0);
                    if (mhz < a.helper_below_mhz) {
                        t = __hip_atomic_fetch_add(a.queue, (uint32_t)PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    if (t1 - born >= la.standby_ticks) break;
                    uint64_t t2;
                    do { // ~50 us asleep: s_sleep and the real-time counter only, no memory traffic
                        __builtin_amdgcn_s_sleep(127);
                        t2 = wall_clock64();
                    } while (t2 - t1 < 5000);
                }
            }
            q_next[0] = t;
        }
        __syncthreads();
        t = q_next[0];
        __syncthreads(); // (the loop below writes q_next[0] again, two trips in)
        active = t != 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NB; ++i) cq[i] = (uint32_t)PREFIX * Gm + t + (uint32_t)i;
        last_static = (uint32_t)PREFIX * Gm + t + (uint32_t)NB;
    }
    bool publish = false; // the first trip has no ticket to publish yet
    if (active && cq[0] < limit_main) {
        u32x4 d[NB][U];
        using W0 = std::integral_constant<int, 0>;
        using WH = std::integral_constant<int, U / 2>;
        using WU = std::integral_constant<int, U>;
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) load(d[i], cq[i]);
        if constexpr (LSP == 2) load_some(d[DEPTH % NB], cq[DEPTH], W0{}, WH{}); // position 1's first half: nobody else will issue it
        bool finished = false;
        while (!finished) {
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                if constexpr (B1 != 0) __builtin_amdgcn_s_barrier();
                if constexpr (TK != 0) { // the ticket for the NEXT trip's load burst: fetched now, published in this trip
                    if (tid == 0) pending = __hip_atomic_fetch_add(a.queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    publish = true;
                }
                if constexpr (LSP == 0) load(d[(p + DEPTH) % NB], cq[DEPTH]);
                else if constexpr (LSP == 1) load_some(d[(p + DEPTH) % NB], cq[DEPTH], W0{}, WH{});
                else load_some(d[(p + DEPTH) % NB], cq[DEPTH], WH{}, WU{});
                __builtin_amdgcn_sched_barrier(0);
                process_store(d[p], cq[0], publish, [&] {
                    if constexpr (LSP == 1) load_some(d[(p + DEPTH) % NB], cq[DEPTH], WH{}, WU{});
                });
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) cq[i] = cq[i + 1];
                cq[DEPTH] = publish ? take_published() : last_static;
                publish = true;
                if (cq[0] >= limit_main) {
                    finished = true;
                    break;
                }
                // LSP 2: chunk k+2 is known now and the buffer it will land in (this trip's) has just been handed to the store burst
                if constexpr (LSP == 2) load_some(d[p], cq[DEPTH], W0{}, WH{});
            }
        }
    }
    if constexpr (TLOOP != 0) {
        // ---- the cold loop over the tail's pieces.  Positions >= n_full are pieces: h = position - n_full is piece (h mod 2^TSPLIT)
        // of chunk n_full + (h >> TSPLIT).  cq[], `pending`, `publish`, the mailbox parity (`trip`) carry over from the trip loop:
        // the protocol simply goes on, with W = U >> TSPLIT words per lane and trip.
        constexpr int W = U >> TSPLIT;
        constexpr uint32_t PIECE = CHUNK >> TSPLIT;
        View wl{nullptr, 0, 0, 0, 0, 1}, ws{nullptr, 0, 0, 0, 0, 1};
        auto locate_t = [&](View &v) { // (one part)
            if (v.origin) return;
            const CycleQueuePart &P = a.part[0];
            v.origin = P.body - P.lead;
            v.end = P.end;
            v.first = P.lead != 0 ? 1u : 0u;
            v.lo = 0;
            v.hi = total;
            v.lane_base = mulmod_canon(P.base_body, lane_mul);
        };
        auto where_t = [&](uint32_t g, const View &v, uint32_t *c, uint32_t *piece) {
            const uint32_t h = g - n_full;
            *c = v.first + n_full + (h >> TSPLIT);
            *piece = h & ((1u << TSPLIT) - 1u);
        };
        auto rsrc_t = [&](uint32_t g, const View &v) {
            uint32_t c, piece;
            where_t(g, v, &c, &piece);
            const uint64_t o = (uint64_t)c * CHUNK + (uint64_t)piece * PIECE;
            const uint64_t left = g < total && o < v.end ? v.end - o : 0;
            return __builtin_amdgcn_make_buffer_rsrc(v.origin + o, 0, (int)(left < PIECE ? left : PIECE), 0x00020000);
        };
        auto load_t = [&](u32x4(&d)[W], uint32_t g) {
            locate_t(wl);
            auto r = rsrc_t(g, wl);
#pragma unroll
            for (int u = 0; u < W; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * SUB, 0, LAUX);
        };
        auto process_store_t = [&](u32x4(&d)[W], uint32_t g, bool pub) {
            locate_t(ws);
            auto r = rsrc_t(g, ws);
            uint32_t c, piece, s[W];
            where_t(g, ws, &c, &piece);
            uint32_t pw = mulmod_canon(c_chunk_pow0<CHUNK>.v[c & 255], c_chunk_pow1<CHUNK>.v[(c >> 8) & 255]);
            pw = mulmod_canon(pw, c_chunk_pow2<CHUNK>.v[(c >> 16) & 255]);
            pw = mulmod_canon(pw, c_tile_lo.v[piece * (PIECE / 4096u)]); // a^(piece * piece bytes)  (entry 0 is 1)
            s[0] = mulmod_canon(ws.lane_base, pw);
#pragma unroll
            for (int u = 1; u < W; ++u) s[u] = mulmod_canon(s[u - 1], lcg::kTileLo.v[BLOCK / 256]);
#pragma unroll
            for (int u = 0; u < W; ++u) d[u] = lab_cycle_word<ALG>(d[u], s[u]);
            if (pub && tid == 0)
                asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(q_next_lds + 4u * (trip & 1u)), "v"(pending) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (tid == 0) pending = __hip_atomic_fetch_add(a.queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < W; ++u) __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, SAUX);
            ++trip;
            stamp(trip);
        };
        if (active && cq[0] < total) { // (cq[0] >= n_full here: the trip loop has run out, or never had a chunk for this workgroup)
            u32x4 e[2][W];
            load_t(e[0], cq[0]);
            bool finished = false;
#pragma unroll 1
            while (!finished) {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    __builtin_amdgcn_s_barrier();
                    load_t(e[(p + 1) % 2], cq[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    process_store_t(e[p], cq[0], publish);
                    cq[0] = cq[1];
                    cq[1] = publish ? take_published() : last_static;
                    publish = true;
                    if (cq[0] >= total) {
                        finished = true;
                        break;
                    }
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
