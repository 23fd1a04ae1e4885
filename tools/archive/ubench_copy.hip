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


// ---- layout experiments on the 1024-thread / 64 KiB shape ---------------------------------
// MAP 0: chunk = block, grid-stride.  MAP 1: XCD-aware -- blocks with equal (blockIdx % 8) share an
// XCD (observed round-robin dispatch); give each XCD one contiguous eighth of the buffer.
// WAVEC: wave-contiguous -- a wave's U loads are adjacent (U KiB per wave) instead of BLOCK*16 apart.
template <int U, int BLOCK, int MAP, bool WAVEC, int RW>
__global__ __launch_bounds__(BLOCK) void shape_k(uint8_t *src, uint8_t *dst, uint64_t n_bytes, uint32_t *sink)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16;
    const uint64_t trips = n_bytes / TRIP;
    uint64_t t0, t1, step;
    if (MAP == 0) { t0 = blockIdx.x; t1 = trips; step = gridDim.x; }
    else { uint64_t per = trips / 8; uint32_t x = blockIdx.x & 7, j = blockIdx.x >> 3; t0 = x * per + j; t1 = (x + 1) * per; step = gridDim.x >> 3; }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 acc = 0;
    for (uint64_t t = t0; t < t1; t += step) {
        auto rs = __builtin_amdgcn_make_buffer_rsrc(src + t * TRIP, 0, (int)TRIP, 0x00020000);
        auto rd = __builtin_amdgcn_make_buffer_rsrc(dst + t * TRIP, 0, (int)TRIP, 0x00020000);
        u32x4 d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t off = WAVEC ? (wave * U + u) * 1024 + lane * 16 : threadIdx.x * 16 + u * BLOCK * 16;
            if (RW & 1) d[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);
            else d[u] = u32x4{(uint32_t)t, off, 1u, 2u};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t off = WAVEC ? (wave * U + u) * 1024 + lane * 16 : threadIdx.x * 16 + u * BLOCK * 16;
            if (RW & 2) __builtin_amdgcn_raw_buffer_store_b128(~d[u], rd, off, 0, 2);
            else acc ^= d[u];
        }
    }
    if (!(RW & 2) && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
}


// write-only / read-only policy sweep on the 1024 x U shape (AUX: 1 = sc0, 2 = nt, 16 = sc1)
template <int U, int BLOCK, int AUX, bool WRITE>
__global__ __launch_bounds__(BLOCK) void pol_k(uint8_t *buf, uint64_t n_bytes, uint32_t *sink)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16;
    const uint64_t trips = n_bytes / TRIP;
    u32x4 acc = 0;
    for (uint64_t t = blockIdx.x; t < trips; t += gridDim.x) {
        auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t off = threadIdx.x * 16 + u * BLOCK * 16;
            if (WRITE) __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)t, off, 1u, 2u}, r, off, 0, AUX);
            else acc ^= __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
        }
    }
    if (!WRITE && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
}


// mixed in-place read-modify-write on the 1024 x U shape with independent load / store policies
template <int U, int BLOCK, int LA, int SA, bool PIPE>
__global__ __launch_bounds__(BLOCK) void mix_k(uint8_t *buf, uint64_t n_bytes)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16;
    const uint64_t trips = n_bytes / TRIP;
    const uint32_t voff = threadIdx.x * 16;
    uint64_t t = blockIdx.x;
    if (t >= trips) return;
    if (!PIPE) {
        for (; t < trips; t += gridDim.x) {
            auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
            u32x4 d[U];
#pragma unroll
            for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * BLOCK * 16, 0, LA);
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(~d[u], r, voff + u * BLOCK * 16, 0, SA);
        }
    } else {
        u32x4 cur[U], nxt[U];
        {
            auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * BLOCK * 16, 0, LA);
        }
        for (; t < trips; t += gridDim.x) {
            uint64_t tn = t + gridDim.x;
            auto rn = __builtin_amdgcn_make_buffer_rsrc(buf + tn * TRIP, 0, tn < trips ? (int)TRIP : 0, 0x00020000);
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = __builtin_amdgcn_raw_buffer_load_b128(rn, voff + u * BLOCK * 16, 0, LA);
            auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(~cur[u], r, voff + u * BLOCK * 16, 0, SA);
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        }
    }
}


// block-synchronous phases: all waves of a workgroup issue their loads, barrier, then all stores
template <int U, int BLOCK, int SYNC>
__global__ __launch_bounds__(BLOCK) void phase_k(uint8_t *buf, uint64_t n_bytes)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16;
    const uint64_t trips = n_bytes / TRIP;
    const uint32_t voff = threadIdx.x * 16;
    uint64_t t = blockIdx.x;
    if (t >= trips) return;
    u32x4 cur[U], nxt[U];
    {
        auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * BLOCK * 16, 0, 2);
    }
    for (; t < trips; t += gridDim.x) {
        uint64_t tn = t + gridDim.x;
        auto rn = __builtin_amdgcn_make_buffer_rsrc(buf + tn * TRIP, 0, tn < trips ? (int)TRIP : 0, 0x00020000);
        if (SYNC & 1) __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = __builtin_amdgcn_raw_buffer_load_b128(rn, voff + u * BLOCK * 16, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (SYNC & 2) __builtin_amdgcn_s_barrier();
        auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(~cur[u], r, voff + u * BLOCK * 16, 0, 16);
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}


// non-pipelined block-synchronous bursts: barrier, U loads, barrier, U stores (two workgroups per CU interleave)
template <int U, int BLOCK>
__global__ __launch_bounds__(BLOCK) void burst_k(uint8_t *buf, uint64_t n_bytes)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16;
    const uint64_t trips = n_bytes / TRIP;
    const uint32_t voff = threadIdx.x * 16;
    for (uint64_t t = blockIdx.x; t < trips; t += gridDim.x) {
        auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
        u32x4 d[U];
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = __builtin_amdgcn_raw_buffer_load_b128(r, voff + u * BLOCK * 16, 0, 2);
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = ~d[u];
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(d[u], r, voff + u * BLOCK * 16, 0, 16);
    }
}


// read-only / write-only with workgroup-synchronous bursts
template <int U, int BLOCK, int AUX, bool WRITE, bool SYNC>
__global__ __launch_bounds__(BLOCK) void polb_k(uint8_t *buf, uint64_t n_bytes, uint32_t *sink)
{
    constexpr uint64_t TRIP = (uint64_t)U * BLOCK * 16;
    const uint64_t trips = n_bytes / TRIP;
    u32x4 acc = 0;
    for (uint64_t t = blockIdx.x; t < trips; t += gridDim.x) {
        auto r = __builtin_amdgcn_make_buffer_rsrc(buf + t * TRIP, 0, (int)TRIP, 0x00020000);
        if (SYNC) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t off = threadIdx.x * 16 + u * BLOCK * 16;
            if (WRITE) __builtin_amdgcn_raw_buffer_store_b128(u32x4{(uint32_t)t, off, 1u, 2u}, r, off, 0, AUX);
            else acc ^= __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX);
        }
    }
    if (!WRITE && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345) sink[0] = 1;
}

struct V { std::string name; void (*fn)(uint8_t *, uint8_t *, uint64_t, uint32_t, hipStream_t); uint32_t grid; bool inplace; double bytes_factor; std::vector<float> ms; };

template <int U, int BLOCK, int LA, int SA, bool CONTIG> void L(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((copy_k<U, BLOCK, LA, SA, CONTIG>), dim3(g), dim3(BLOCK), 0, st, s, d, n); }
template <int U, int BLOCK, int LA, int SA> void LP(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((copy_pipe_k<U, BLOCK, LA, SA>), dim3(g), dim3(BLOCK), 0, st, s, d, n); }
static uint32_t *g_sink;
template <int U, int BLOCK, int MAP, bool WAVEC, int RW> void LS(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((shape_k<U, BLOCK, MAP, WAVEC, RW>), dim3(g), dim3(BLOCK), 0, st, s, d, n, g_sink); }
template <int U, int BLOCK, int AUX, bool WRITE> void LPOL(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((pol_k<U, BLOCK, AUX, WRITE>), dim3(g), dim3(BLOCK), 0, st, s, n, g_sink); }
template <int U, int BLOCK, int LA, int SA, bool PIPE> void LMIX(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((mix_k<U, BLOCK, LA, SA, PIPE>), dim3(g), dim3(BLOCK), 0, st, s, n); }
template <int U, int BLOCK, int SYNC> void LPH(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((phase_k<U, BLOCK, SYNC>), dim3(g), dim3(BLOCK), 0, st, s, n); }
template <int U, int BLOCK> void LBU(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((burst_k<U, BLOCK>), dim3(g), dim3(BLOCK), 0, st, s, n); }
template <int U, int BLOCK, int AUX, bool WRITE, bool SYNC> void LPB(uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st)
{ hipLaunchKernelGGL((polb_k<U, BLOCK, AUX, WRITE, SYNC>), dim3(g), dim3(BLOCK), 0, st, s, n, g_sink); }
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
    CHECK(hipMalloc(&g_sink, 64));
    std::vector<V> vs;
    auto add = [&](const char *nm, decltype(V::fn) fn, uint32_t g, bool inpl, double f = 2.0) {
        char buf[128]; snprintf(buf, sizeof buf, "%-34s grid=%5u %s", nm, g, inpl ? "in-place " : "out-of-pl");
        vs.push_back({buf, fn, g, inpl, f, {}});
    };
    CHECK(hipMalloc(&g_sink, 64));
    add("hipMemset (write ref)", [](uint8_t *s, uint8_t *d, uint64_t n, uint32_t g, hipStream_t st) { (void)hipMemsetAsync(s, 0x33, n, st); }, 0, true, 1.0);
    for (uint32_t g : {256u, 512u, 1024u, 2048u}) {
        add("write U4 sc1 nosync", LPB<4, 1024, 16, true, false>, g, true, 1.0);
        add("write U4 sc1 sync", LPB<4, 1024, 16, true, true>, g, true, 1.0);
        add("write U8 sc1 nosync", LPB<8, 1024, 16, true, false>, g, true, 1.0);
        add("write U8 sc1 sync", LPB<8, 1024, 16, true, true>, g, true, 1.0);
        add("write U8 plain sync", LPB<8, 1024, 0, true, true>, g, true, 1.0);
        add("write U8 plain nosync", LPB<8, 1024, 0, true, false>, g, true, 1.0);
        add("write U16 sc1 sync", LPB<16, 1024, 16, true, true>, g, true, 1.0);
        add("write U2 sc1 sync", LPB<2, 1024, 16, true, true>, g, true, 1.0);
        add("write U8 B256 plain", LPB<8, 256, 0, true, false>, g, true, 1.0);
        add("read U4 nt nosync", LPB<4, 1024, 2, false, false>, g, true, 1.0);
        add("read U4 nt sync", LPB<4, 1024, 2, false, true>, g, true, 1.0);
        add("read U8 nt sync", LPB<8, 1024, 2, false, true>, g, true, 1.0);
        add("read U2 nt sync", LPB<2, 1024, 2, false, true>, g, true, 1.0);
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
