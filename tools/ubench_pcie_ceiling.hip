// tools/ubench_pcie_ceiling.hip -- what does the PCIe link give, measured by things the library under test did not write?
//
// the host-buffer routes' roofline used to be priced against "the best one kernel of ours does in place on
// page-locked memory" (50.4 GB/s) -- a tautology for the pinned route.  This prints, in ONE run on 1 GiB of page-locked memory:
//   * the DMA engines (hipMemcpyAsync, 16 MiB pieces): H2D alone, D2H alone, both at once (per direction);
//   * a kernel across PCIe in the product's small shape (256 lanes, one 16-byte word per lane per trip, loads nt, stores sc1),
//     READ-ONLY (every word loaded and folded into a register), WRITE-ONLY (the keystream stored, nothing loaded), the COPY in
//     place (load + store, no arithmetic) and IN PLACE with the arithmetic -- each with 32 and 256 workgroups;
//   * the product itself on the same pages (modgpu_cycle_host on page-locked memory = the pinned route).
// If read-only and write-only each reach the DMA engines' one-way rate and the mix does not, what is missing is the link's cost of
// running both directions at once; if reads alone fall short, it is the request size the 1 KiB wave loads become on the link.
// The last line is JSON (tools/summarize_pcie_trace.py --ceilings reads it: every roofline_pcie object carries these figures).
//
//   tools/ubench_pcie_ceiling [MiB = 1024] [reps = 5]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/modgpu.h"
#include "../include/modgpu_testing.h"
#include "cycle_kernel_impl.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

enum Mode { READ_ONLY = 0, WRITE_ONLY = 1, COPY = 2, CYCLE = 3 };

// the small shape's trip: lane t of workgroup g owns word g * 256 + t of every grid-sized stripe
template <int MODE> __global__ __launch_bounds__(256) void pcie_kernel(uint8_t *p, uint64_t words, uint32_t base, uint32_t stride_mul2, uint32_t *sink)
{
    const uint32_t tid = threadIdx.x;
    auto r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)std::min<uint64_t>(words * 16, 0x7FFFFFF0ull), 0x00020000);
    uint32_t s = mulmod_canon(base, c_lane_pow.v[tid]);
    s = mulmod_canon(s, lcg::powmod(lcg::A, 0)); // (kept simple: the keystream's POSITION is not what is measured here)
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t w = (uint64_t)blockIdx.x * 256 + tid; w < words; w += (uint64_t)gridDim.x * 256) {
        const uint32_t o = (uint32_t)(w * 16);
        u32x4 d = {0, 0, 0, 0};
        if (MODE != WRITE_ONLY) d = __builtin_amdgcn_raw_buffer_load_b128(r, o, 0, AUX_NT);
        if (MODE == CYCLE || MODE == WRITE_ONLY) d = cycle_word<1>(d, s);
        if (MODE == READ_ONLY) acc ^= d;
        else __builtin_amdgcn_raw_buffer_store_b128(d, r, o, 0, AUX_SC1);
        if (MODE == CYCLE || MODE == WRITE_ONLY) s = mulmod_canon2(s, stride_mul2);
    }
    if (MODE == READ_ONLY) sink[blockIdx.x * 256 + tid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

struct Row { const char *name; double best, median; };

template <typename F> static Row timed(const char *name, uint64_t bytes, int reps, F &&run)
{
    run(); // warm
    std::vector<double> v;
    for (int r = 0; r < reps; ++r) {
        const double t0 = now();
        run();
        v.push_back(bytes / (now() - t0) / 1e9);
    }
    std::sort(v.begin(), v.end());
    return {name, v.back(), v[v.size() / 2]};
}

int main(int argc, char **argv)
{
    const uint64_t n = (argc > 1 ? strtoull(argv[1], nullptr, 0) : 1024ull) << 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    if (n >= (2ull << 30)) { printf("at most 2047 MiB (one buffer descriptor)\n"); return 1; }
    const uint64_t chunk = 16ull << 20;
    uint8_t *h0 = nullptr, *h1 = nullptr, *d0, *d1;
    // modgpu_host_alloc: page-locked and known to the library as such (the product row takes the pinned route on it)
    if (modgpu_host_alloc((void **)&h0, n) != MODGPU_OK || !modgpu_host_is_pinned(h0, n)) { printf("modgpu_host_alloc: %s\n", modgpu_last_error()); return 1; }
    CHECK(hipHostMalloc((void **)&h1, n, hipHostMallocDefault));
    memset(h0, 1, n); memset(h1, 2, n);
    CHECK(hipMalloc((void **)&d0, n)); CHECK(hipMalloc((void **)&d1, n));
    uint32_t *sink; CHECK(hipMalloc((void **)&sink, 256 * 256 * 4));
    hipStream_t s0, s1; CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    std::vector<Row> rows;
    for (int mode = 0; mode < 3; ++mode)
        rows.push_back(timed(mode == 0 ? "dma_h2d" : mode == 1 ? "dma_d2h" : "dma_duplex_per_direction", n, reps, [&] {
            for (uint64_t o = 0; o < n; o += chunk) {
                const uint64_t l = std::min(chunk, n - o);
                if (mode != 1) CHECK(hipMemcpyAsync(d0 + o, h0 + o, l, hipMemcpyHostToDevice, s0));
                if (mode != 0) CHECK(hipMemcpyAsync(h1 + o, d1 + o, l, hipMemcpyDeviceToHost, s1));
            }
            CHECK(hipStreamSynchronize(s0)); CHECK(hipStreamSynchronize(s1));
        }));
    uint8_t *mapped; CHECK(hipHostGetDevicePointer((void **)&mapped, h0, 0));
    const uint32_t base = lcg::state_residue(lcg::key_residue((int32_t)0x90cfc0ab), 0);
    static char names[16][48];
    int k = 0;
    for (uint32_t grid : {32u, 256u}) {
        const uint32_t stride2 = 2u * lcg::powmod(lcg::A, ((uint64_t)grid * 4096) % lcg::PERIOD);
        for (int mode = 0; mode < 4; ++mode) {
            snprintf(names[k], sizeof names[k], "kernel_%s_grid%u", mode == READ_ONLY ? "read_only" : mode == WRITE_ONLY ? "write_only" : mode == COPY ? "copy_in_place" : "cycle_in_place", grid);
            rows.push_back(timed(names[k++], n, reps, [&] {
                switch (mode) {
                case READ_ONLY: hipLaunchKernelGGL(pcie_kernel<READ_ONLY>, dim3(grid), dim3(256), 0, s0, mapped, n / 16, base, stride2, sink); break;
                case WRITE_ONLY: hipLaunchKernelGGL(pcie_kernel<WRITE_ONLY>, dim3(grid), dim3(256), 0, s0, mapped, n / 16, base, stride2, sink); break;
                case COPY: hipLaunchKernelGGL(pcie_kernel<COPY>, dim3(grid), dim3(256), 0, s0, mapped, n / 16, base, stride2, sink); break;
                default: hipLaunchKernelGGL(pcie_kernel<CYCLE>, dim3(grid), dim3(256), 0, s0, mapped, n / 16, base, stride2, sink); break;
                }
                CHECK(hipStreamSynchronize(s0));
            }));
        }
    }
    rows.push_back(timed("product_pinned_route_in_place", n, reps, [&] {
        if (modgpu_cycle_host(h0, n, (int32_t)0x90cfc0ab, 0, 0) != MODGPU_OK) { printf("modgpu_cycle_host: %s\n", modgpu_last_error()); exit(1); }
    }));
    modgpu_launch_info_t ll{};
    modgpu_last_launch(&ll);
    printf("== PCIe ceilings on this node: %llu MiB of page-locked memory, %d repetitions, GB/s of payload (= per direction), best / median\n", (unsigned long long)(n >> 20), reps);
    printf("   gen 5 x16 on paper: 64 GB/s per direction (63.0 after 128b/130b)\n");
    for (const Row &r : rows) printf("   %-36s %7.2f %7.2f\n", r.name, r.best, r.median);
    printf("   (product row: %s, grid %u)\n", ll.kernel ? ll.kernel : "?", ll.grid);
    printf("CEILING {\"MiB\": %llu, \"peak_link\": 64.0", (unsigned long long)(n >> 20));
    for (const Row &r : rows) printf(", \"%s\": %.2f", r.name, r.median);
    printf(", \"kernel_source_hash\": \"%s\", \"feed_kernel_source_hash\": \"%s\"}\n", modgpu_kernel_source_hash(), modgpu_feed_kernel_source_hash());
    return 0;
}
