// tools/ubench_queue_rw.hip -- what the memory system gives the work-queue access pattern when it only reads, only
// writes, or copies in place: the ceilings the cycle kernel (read + compute + write in place) sits under.
// Same schedule as modgpu_cycle_queue_kernel (64 KiB chunks, 1024-thread workgroups, static prefix of 2 + tickets
// fetched at the start of the trip before the one that loads them, nt loads, sc1 nt stores, both workgroup barriers), no arithmetic.
// (profiles/r04_memside_counters.json and r02_ubench_queue_rw.txt were taken with round 3's ticket timing -- prefix of 3, the
//  ticket published a trip later -- in all four kernels.)
// Round 4: the fourth row is the PRODUCT kernel itself (modgpu_cycle_queue_kernel<4, 1024> from the product header), so that one
// `rocprofv3 --pmc` pass over this program reads the memory-side counters of all four under the same conditions
// (tools/memside_counters.sh, profiles/r04_memside_counters.json).
// Round 5: two more rows, the lab kernel at the product's settings with the next chunk's loads SPREAD over the trip (LSP 1 / 2,
// tools/cycle_kernel_lab.h), so that the same counter pass says what that does to the reads in flight (profiles/r05_lsp.txt).
// Build: make -C tools ubench_queue_rw
// Run:   tools/ubench_queue_rw [bytes=4294967296] [grid=200] [rounds=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "cycle_kernel_lab.h" // (includes the product header; the lab kernel only for round 5's LSP rows)
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
enum { READ = 0, WRITE = 1, COPY = 2, FULL = 3 };
constexpr int U = 4, BLOCK = 1024;
constexpr uint32_t SUB = BLOCK * 16, CHUNK = U * SUB;

template <int MODE> __global__ __launch_bounds__(BLOCK) void rw_kernel(uint8_t *buf, uint64_t bytes, uint32_t *queue, uint32_t *sink)
{
    const uint32_t tid = threadIdx.x, blk = blockIdx.x, G = gridDim.x;
    __shared__ uint32_t q_next[2];
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&q_next[0];
    const uint32_t n_chunks = (uint32_t)((bytes + CHUNK - 1) / CHUNK), voff = tid * 16, one = 1u;
    uint32_t trip = 0, pending = 0;
    u32x4 acc = {0, 0, 0, 0};
    auto rsrc_at = [&](uint32_t c) {
        const uint64_t o = (uint64_t)c * CHUNK, left = o < bytes ? bytes - o : 0;
        return __builtin_amdgcn_make_buffer_rsrc(buf + o, 0, (int)(left < CHUNK ? left : CHUNK), 0x00020000);
    };
    auto load = [&](u32x4(&d)[U], uint32_t c) {
        if constexpr (MODE == WRITE) {
            for (int u = 0; u < U; ++u) d[u] = u32x4{tid, c, (uint32_t)u, trip};
        } else {
            auto r = rsrc_at(c);
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * SUB, 0, 2);
        }
    };
    auto process_store = [&](u32x4(&d)[U], uint32_t c) {
        auto r = rsrc_at(c);
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = ~d[u];
        if (tid == 0) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(lds + 4u * (trip & 1u)), "v"(pending) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (MODE == READ) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= d[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * SUB, 0, 18);
        }
        ++trip;
    };
    auto take = [&]() {
        uint32_t t;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lds + 4u * ((trip - 1u) & 1u)) : "memory");
        return 2u * G + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    };
    uint32_t c0 = blk, c1 = blk + G;
    auto fetch = [&]() { if (tid == 0) pending = __hip_atomic_fetch_add(queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (c0 < n_chunks) {
        u32x4 d0[U], d1[U];
        load(d0, c0);
        while (true) {
            __builtin_amdgcn_s_barrier();
            fetch();
            load(d1, c1);
            __builtin_amdgcn_sched_barrier(0);
            process_store(d0, c0);
            c0 = c1;
            c1 = take();
            if (c0 >= n_chunks) break;
            __builtin_amdgcn_s_barrier();
            fetch();
            load(d0, c1);
            __builtin_amdgcn_sched_barrier(0);
            process_store(d1, c0);
            c0 = c1;
            c1 = take();
            if (c0 >= n_chunks) break;
        }
    }
    if constexpr (MODE == READ) {
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[blk] = acc.x; // keeps the loads alive
    }
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(queue + 1, 1u) == G - 1) {
            __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 32);
    uint32_t grid = argc > 2 ? (uint32_t)atoi(argv[2]) : 200u;
    const int rounds = argc > 3 ? atoi(argv[3]) : 8;
    uint8_t *buf;
    uint32_t *queue, *sink;
    CHECK(hipMalloc(&buf, n));
    CHECK(hipMemset(buf, 0x5A, n));
    CHECK(hipMalloc(&queue, 256));
    CHECK(hipMemset(queue, 0, 256));
    CHECK(hipMalloc(&sink, 4096 * 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[6] = {"read-only ", "write-only", "copy r+w  ", "PRODUCT   ", "LAB LSP 1 ", "LAB LSP 2 "};
    std::vector<float> ms[6];
    // the product kernel's table of one part (the buffer is chunk-aligned: no lead, no edges)
    CycleQueueArgs qa{};
    qa.queue = queue + 32; // a line of its own
    qa.n_parts = 1;
    qa.part[0].body = buf;
    qa.part[0].end = n / 16 * 16;
    qa.part[0].base_body = qa.part[0].base_head = qa.part[0].base_tail = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    for (int k = 1; k <= kCycleBatchMax; ++k) qa.start[k] = (uint32_t)((qa.part[0].end + CHUNK - 1) / CHUNK);
    const LabQueueArgs lqa{qa, nullptr, 0};
    for (int r = 0; r < rounds; ++r)
        for (int m = 0; m < 6; ++m) {
            CHECK(hipEventRecord(e0, st));
            for (int k = 0; k < 2; ++k) {
                if (m == READ) hipLaunchKernelGGL(rw_kernel<READ>, dim3(grid), dim3(BLOCK), 0, st, buf, n, queue, sink);
                else if (m == WRITE) hipLaunchKernelGGL(rw_kernel<WRITE>, dim3(grid), dim3(BLOCK), 0, st, buf, n, queue, sink);
                else if (m == COPY) hipLaunchKernelGGL(rw_kernel<COPY>, dim3(grid), dim3(BLOCK), 0, st, buf, n, queue, sink);
                else if (m == FULL) hipLaunchKernelGGL((modgpu_cycle_queue_kernel<U, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, qa);
                else if (m == 4) hipLaunchKernelGGL((lab_cycle_queue_kernel<U, BLOCK, 2, 18, 0, 1, MODE_FULL, 2, 1, 1, 0, 1, 0, 1, 0>), dim3(grid), dim3(BLOCK), 0, st, lqa);
                else hipLaunchKernelGGL((lab_cycle_queue_kernel<U, BLOCK, 2, 18, 0, 1, MODE_FULL, 2, 1, 1, 0, 1, 0, 2, 0>), dim3(grid), dim3(BLOCK), 0, st, lqa);
            }
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            if (r >= 2 || rounds < 3) ms[m].push_back(t / 2);
        }
    CHECK(hipGetLastError());
    printf("bytes=%llu grid=%u, work-queue schedule, 64 KiB chunks (GB/s of HBM traffic: n for read-only / write-only, 2n for copy)\n", (unsigned long long)n, grid);
    for (int m = 0; m < 6; ++m) {
        std::sort(ms[m].begin(), ms[m].end());
        const double traffic = (m >= COPY ? 2.0 : 1.0) * n;
        printf("  %s  median %.4f ms -> %7.1f GB/s   best %7.1f\n", names[m], ms[m][ms[m].size() / 2], traffic / ms[m][ms[m].size() / 2] / 1e6,
               traffic / ms[m].front() / 1e6);
    }
    return 0;
}
