// tools/ubench_copy.hip -- what does the MI355X memory system give an IN-PLACE read-modify-write
// byte stream (the cycle kernel's access pattern), and which cache-policy bits / grid shapes /
// in-flight depths get closest to it?  Every variant reads 16 B per lane, flips the bits, and
// stores 16 B per lane; GB/s = (bytes read + bytes written) / time.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_copy.hip -o tools/ubench_copy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

using u32x4 = uint32_t __attribute__((ext_vector_type(4)));

// LA / SA: aux cache-policy bits for loads / stores (1 = sc0, 2 = nt, 16 = sc1)
// CONTIG: each block walks its own contiguous chunk; else grid-stride over tiles
template <int U, int BLOCK, int LA, int SA, bool CONTIG>
__global__ __launch_bounds__(BLOCK) void copy_k(uint8_t *src, uint8_t *dst, uint64_t n_bytes)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16; // bytes per block per trip
    const uint64_t trips = n_bytes / TRIP;             // n_bytes is a multiple of TRIP*grid
    uint64_t t0, t1, step;
    if (CONTIG) { uint64_t per = trips / gridDim.x; t0 = blockIdx.x * per; t1 = t0 + per; step = 1; }
    else { t0 = blockIdx.x; t1 = trips; step = gridDim.x; }
    const uint32_t voff = threadIdx.x * 16;
    for (uint64_t t = t0; t < t1; t += step) {
        auto rs = __builtin_amdgcn_make_buffer_rsrc(src + t * TRIP, 0, (int)TRIP, 0x00020000);
        auto rd = __builtin_amdgcn_make_buffer_rsrc(dst + t * TRIP, 0, (int)TRIP, 0x00020000);
        u32x4 d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + u * BLOCK * 16, 0, LA);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(~d[u], rd, voff + u * BLOCK * 16, 0, SA);
    }
}

// software-pipelined: loads of trip k+1 are issued before the stores of trip k
template <int U, int BLOCK, int LA, int SA>
__global__ __launch_bounds__(BLOCK) void copy_pipe_k(uint8_t *src, uint8_t *dst, uint64_t n_bytes)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16;
    const uint64_t trips = n_bytes / TRIP;
    const uint32_t voff = threadIdx.x * 16;
    uint64_t t = blockIdx.x;
    if (t >= trips) return;
    u32x4 cur[U], nxt[U];
    {
        auto rs = __builtin_amdgcn_make_buffer_rsrc(src + t * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + u * BLOCK * 16, 0, LA);
    }
    for (; t < trips; t += gridDim.x) {
        uint64_t tn = t + gridDim.x;
        if (tn < trips) {
            auto rs = __builtin_amdgcn_make_buffer_rsrc(src + tn * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + u * BLOCK * 16, 0, LA);
        }
        auto rd = __builtin_amdgcn_make_buffer_rsrc(dst + t * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(~cur[u], rd, voff + u * BLOCK * 16, 0, SA);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}

__global__ void read_k(const u32x4 *src, uint64_t n_words, uint32_t *sink)
{
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, st = (uint64_t)gridDim.x * blockDim.x;
    u32x4 acc = 0;
    for (; i + 3 * st < n_words; i += 4 * st) {
        u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + st);
        u32x4 c = __builtin_nontemporal_load(src + i + 2 * st), d = __builtin_nontemporal_load(src + i + 3 * st);
        acc ^= a ^ b ^ c ^ d;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
}

__global__ void write_k(u32x4 *dst, uint64_t n_words)
{
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, st = (uint64_t)gridDim.x * blockDim.x;
    u32x4 v = {1u, 2u, 3u, (uint32_t)i};
    for (; i < n_words; i += st) __builtin_nontemporal_store(v, dst + i);
}

struct V { std::string name; void (*fn)(uint8_t *, uint8_t *, uint64_t, uint32_t, hipStream_t); uint32_t grid; bool inplace; double bytes_factor; std::vector<float> ms; };

template <int U, int BLOCK, int LA, int SA, bool CONTIG> void L(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((copy_k<U, BLOCK, LA, SA, CONTIG>), dim3(g), dim3(BLOCK), 0, st, s, d, n); }
template <int U, int BLOCK, int LA, int SA> void LP(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((copy_pipe_k<U, BLOCK, LA, SA>), dim3(g), dim3(BLOCK), 0, st, s, d, n); }
void LR(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st) { hipLaunchKernelGGL(read_k, dim3(g), dim3(256), 0, st, (const u32x4 *)s, n / 16, (uint32_t *)d); }
void LW(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st) { hipLaunchKernelGGL(write_k, dim3(g), dim3(256), 0, st, (u32x4 *)s, n / 16); }
void LM(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st) { (void)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st); }

int main(int argc, char **argv)
{
    uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 32);
    int rounds = argc > 2 ? atoi(argv[2]) : 5;
    uint8_t *a, *b;
    CHECK(hipMalloc(&a, n)); CHECK(hipMalloc(&b, n));
    CHECK(hipMemset(a, 0x5A, n)); CHECK(hipMemset(b, 0x11, n));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<V> vs;
    auto add = [&](const char *nm, decltype(V::fn) fn, uint32_t g, bool inpl, double f = 2.0) {
        char buf[128]; snprintf(buf, sizeof buf, "%-34s grid=%5u %s", nm, g, inpl ? "in-place " : "out-of-pl");
        vs.push_back({buf, fn, g, inpl, f, {}});
    };
    add("read-only nt U4", LR, 2048, true, 1.0);
    add("write-only nt", LW, 2048, true, 1.0);
    add("hipMemcpyDtoD", LM, 0, false);
    for (bool inpl : {true, false}) {
        for (uint32_t g : {256u, 512u, 1024u, 2048u}) {
            add("U4 B256 ld=nt st=nt stride", L<4, 256, 2, 2, false>, g, inpl);
            add("U8 B256 ld=nt st=nt stride", L<8, 256, 2, 2, false>, g, inpl);
            add("U4 B256 ld=nt st=nt contig", L<4, 256, 2, 2, true>, g, inpl);
            add("U4 B256 ld=0  st=0  stride", L<4, 256, 0, 0, false>, g, inpl);
            add("U4 B256 ld=nt st=0  stride", L<4, 256, 2, 0, false>, g, inpl);
            add("U4 B256 ld=0  st=nt stride", L<4, 256, 0, 2, false>, g, inpl);
            add("U4 B256 ld=sc1 st=sc1 stride", L<4, 256, 16, 16, false>, g, inpl);
            add("U4 B256 ld=sc0sc1 st=sc0sc1", L<4, 256, 17, 17, false>, g, inpl);
            add("U4 B256 ld=nt+sc1 st=nt+sc1", L<4, 256, 18, 18, false>, g, inpl);
            add("U4 B256 ld=all st=all", L<4, 256, 19, 19, false>, g, inpl);
            add("U4 B256 pipe ld=nt st=nt", LP<4, 256, 2, 2>, g, inpl);
            add("U8 B256 pipe ld=nt st=nt", LP<8, 256, 2, 2>, g, inpl);
        }
        for (uint32_t g : {256u, 512u, 1024u}) {
            add("U4 B512 ld=nt st=nt stride", L<4, 512, 2, 2, false>, g, inpl);
            add("U2 B1024 ld=nt st=nt stride", L<2, 1024, 2, 2, false>, g, inpl);
            add("U4 B1024 ld=nt st=nt stride", L<4, 1024, 2, 2, false>, g, inpl);
            add("U4 B512 pipe ld=nt st=nt", LP<4, 512, 2, 2>, g, inpl);
        }
    }
    for (int r = 0; r < rounds + 1; ++r)
        for (auto &v : vs) {
            CHECK(hipEventRecord(e0, st));
            v.fn(a, v.inplace ? a : b, n, v.grid, st);
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r) v.ms.push_back(ms);
        }
    CHECK(hipGetLastError());
    printf("bytes=%llu rounds=%d\n", (unsigned long long)n, rounds);
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        float med = v.ms[v.ms.size() / 2];
        printf("%s  med %.4f ms min %.4f -> %7.1f GB/s\n", v.name.c_str(), med, v.ms.front(), v.bytes_factor * n / med / 1e6);
    }
    return 0;
}
