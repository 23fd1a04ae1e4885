// cycle_feed_kernel.hip -- the host-fed kernel of the pageable host route (see cycle_feed_kernel.h).  Its own TU: the arithmetic
// is cycle_kernel_impl.h's (cycle_word<1>, the Mersenne-fold multiply, the jump tables -- the small shape's, which is what runs
// across PCIe), the loop is new.  PCIe-bound byte work: one 16-byte word per lane per trip, loads `nt`, stores `sc1`, 32
// workgroups of 256 lanes (more only contend for the link: modgpu_capi.cpp kPcieGridShort, profiles/r05_pcie_persist.txt).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "cycle_feed_kernel.h"
#include "cycle_kernel_impl.h"

namespace {
constexpr uint32_t kNone = 0xFFFFFFFFu;
} // namespace

// Shape of the loop: everything only thread 0 does sits in ONE region at the top of a trip, in front of the first barrier --
// counting the piece of the trip before, then drawing the next ticket and waiting for its chunk.  (A second thread-0 region
// behind the trip's last barrier had the compiler send lanes 1..63 of wave 0 round the back edge on their own: that wave then
// passed the barrier twice per trip, the other waves once, and the workgroup hung -- found in the lab form of this kernel,
// tools/archive/ubench_pcie_persist.hip.)  Two things keep that from coming back with a compiler's mood: what decides whether a wave
// goes round again (the ticket, the ok word) is read out of LDS into SCALAR registers (readfirstlane), so the back edge is a
// scalar branch that a wave takes whole or not at all; and check_isa.py follows the compiled kernel's control flow and refuses
// a build in which either s_barrier can be reached with anything but the EXEC mask the wave entered the loop with.
//
// Leaving: a workgroup's wait for a chunk ends when the host raises `abort`, when its patience runs out, or when ANOTHER
// workgroup has already given up (work[1] != 0): the tickets that one held are lost, the call cannot be completed by this
// kernel any more, and everybody who would otherwise wait out a patience of their own leaves at once -- the host starts its
// rescue one patience after it went away, not two.
__global__ __launch_bounds__(256) void modgpu_cycle_feed_kernel(CycleFeedArgs a)
{
    __shared__ uint32_t s_t, s_ok;
    const uint32_t tid = threadIdx.x;
    const uint32_t tpc = a.chunk_bytes / kFeedPieceBytes;
    const uint32_t n_tickets = (uint32_t)((a.n + kFeedPieceBytes - 1) / kFeedPieceBytes);
    const uint32_t n_chunks = (n_tickets + tpc - 1) / tpc;
    uint32_t counted = kNone; // thread 0: the chunk of the piece this workgroup has finished and not yet counted
    for (;;) {
        if (tid == 0) {
            if (counted != kNone) {
                const uint32_t pieces = counted + 1 < n_chunks ? tpc : n_tickets - counted * tpc;
                if (__hip_atomic_fetch_add(&a.work[2 + counted], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == pieces - 1)
                    __hip_atomic_store(&a.done[counted], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                counted = kNone;
            }
            const uint32_t t = atomicAdd(&a.work[0], 1u);
            uint32_t ok = 1;
            if (t < n_tickets) {
                const uint32_t c = t / tpc;
                counted = c;
                const uint64_t since = wall_clock64();
                while (__hip_atomic_load(&a.ready[c], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
                    // the exits every waiting wave reaches: the host gives the call up, does not turn up at all, or a sibling has left already
                    if (__hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u || wall_clock64() - since > a.patience_ticks ||
                        __hip_atomic_load(&a.work[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(16);
                }
            }
            s_t = t;
            s_ok = ok;
        }
        __syncthreads();
        // (the same value in every lane: held in scalar registers, the trip's control flow below is the whole wave's)
        const uint32_t t = __builtin_amdgcn_readfirstlane(s_t), ok = __builtin_amdgcn_readfirstlane(s_ok);
        if (t >= n_tickets || !ok) {
            if (!ok && tid == 0) atomicAdd(&a.work[1], 1u);
            break;
        }
        const uint32_t c = t / tpc, piece = t - c * tpc;
        const uint64_t pos = (uint64_t)t * kFeedPieceBytes; // stream position of the piece's first byte
        uint8_t *const p = a.slot[(c % a.pipes) * 2u + (c / a.pipes) % 2u] + (uint64_t)piece * kFeedPieceBytes; // the piece in its chunk's staging slot
        const uint32_t len = (uint32_t)(a.n - pos < kFeedPieceBytes ? a.n - pos : kFeedPieceBytes);
        const uint32_t words = len / lcg::WORD;
        // state of the piece's first byte: base * a^(32768 * t), by the three bytes of t
        uint32_t sp = mulmod_canon(a.base, c_chunk_pow0<kFeedPieceBytes>.v[t & 255]);
        sp = mulmod_canon(sp, c_chunk_pow1<kFeedPieceBytes>.v[(t >> 8) & 255]);
        sp = mulmod_canon(sp, c_chunk_pow2<kFeedPieceBytes>.v[t >> 16]);
        uint32_t s = mulmod_canon(sp, c_lane_pow.v[tid]); // this lane's first word
        // the hardware range check (num_records = the piece's whole words) drops the lanes past the end of a short last piece
        auto r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(words * lcg::WORD), 0x00020000);
        const uint32_t trips = (words + 255u) / 256u;
        for (uint32_t j = 0; j < trips; ++j) {
            const uint32_t o = (j * 256u + tid) * lcg::WORD;
            u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, AUX_NT);
            d = cycle_word<1>(d, s);
            __builtin_amdgcn_raw_buffer_store_b128(d, r, o, 0, AUX_SC1);
            s = mulmod_canon(s, lcg::kTileLo.v[1]); // a^4096: the same lane, one trip on
        }
        // < 16 bytes behind the last whole word of the call's last piece, bytewise
        const uint32_t tail = len - words * lcg::WORD;
        if (tid < tail) {
            uint32_t st = mulmod_canon(sp, c_lane_pow.v[words & 255]); // a^(16 * words) = a^(16 * (words % 256)) * a^(4096 * (words / 256))
            st = mulmod_canon(st, c_tile_lo.v[words >> 8]);
            for (uint32_t k = 0; k < tid; ++k) st = mulmod_canon(st, lcg::A);
            uint8_t *const q = p + words * lcg::WORD + tid;
            *q = cycle_byte(*q, st);
        }
        __threadfence_system(); // this wave's stores have reached host memory ...
        __syncthreads();        // ... and every wave's have, before thread 0 counts the piece at the top of the next trip
    }
}

uint32_t modgpu_feed_block() { return 256u; }
const char *modgpu_feed_kernel_name() { return "modgpu_cycle_feed_kernel"; }
hipError_t modgpu_launch_cycle_feed(const CycleFeedArgs &a, uint32_t grid, hipStream_t stream)
{
    hipLaunchKernelGGL(modgpu_cycle_feed_kernel, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}
