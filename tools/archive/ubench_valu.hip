// tools/ubench_valu.hip -- measures the issue cost (cycles per wave64 instruction per SIMD) of the
// integer VALU instructions the cycle kernel is built from, on the GPU it runs on.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o tools/ubench_valu
// Each wave runs REP x 8 independent instances of one instruction and stamps s_memtime around
// the loop; waves/SIMD is swept so both the single-wave issue cost and the saturated rate show.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int REP = 512;

#define OP8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)

template <int OP>
__global__ void k(unsigned long long *cyc, unsigned *sink, unsigned seed)
{
    unsigned a[8], b[8];
    unsigned long long w[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 3 + i); b[i] = seed + i * 77 + threadIdx.x; w[i] = a[i]; }
    unsigned c = seed | 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int r = 0; r < REP; ++r) {
        if constexpr (OP == 0) {
#define S(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 1) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "s"(c) : "vcc");
            OP8(S)
#undef S
        } else if constexpr (OP == 2) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 3) {
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 4) {
#define S(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 5) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 6) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(c));
            OP8(S)
#undef S
        } else if constexpr (OP == 7) {
#define S(i) asm volatile("v_add_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i+1)&7]));
            OP8(S)
#undef S
        } else if constexpr (OP == 8) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 9) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 10) {
#define S(i) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 11) {
#define S(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(c));
            OP8(S)
#undef S
        } else if constexpr (OP == 12) { // v_mad_u64_u32 with a VGPR*VGPR product, dependent on nothing
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            OP8(S)
#undef S
        } else if constexpr (OP == 13) {
#define S(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            OP8(S)
#undef S
        } else if constexpr (OP == 14) {
#define S(i) asm volatile("v_lshrrev_b32 %0, 1, %1" : "=v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 15) {
#define S(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        } else if constexpr (OP == 16) {
#define S(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i+1)&7]));
            OP8(S)
#undef S
        } else if constexpr (OP == 17) {
#define S(i) asm volatile("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(a[i]) : "v"(b[i]), "v"(b[(i+1)&7]));
            OP8(S)
#undef S
        } else if constexpr (OP == 18) {
#define S(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            OP8(S)
#undef S
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + (unsigned)w[i] + (unsigned)(w[i] >> 32);
    if (acc == 0x12345678) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP> void run(const char *name, unsigned long long *d_cyc, unsigned *d_sink)
{
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {
        // blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 12345u);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks * 4);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        double med = (double)h[h.size() / 2];
        // a SIMD hosts wps waves, each issuing REP*8 instructions in `med` cycles
        printf("  w%d: %6.2f cyc/inst/wave -> %5.2f cyc/inst/SIMD", wps, med / (REP * 8.0), med / (REP * 8.0) / wps);
    }
    printf("\n");
}

int main()
{
    unsigned long long *d_cyc; unsigned *d_sink;
    CHECK(hipMalloc(&d_cyc, 8 * 2048 * 4)); CHECK(hipMalloc(&d_sink, 64));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("v_add_u32", d_cyc, d_sink);
    run<10>("v_xnor_b32", d_cyc, d_sink);
    run<14>("v_lshrrev_b32", d_cyc, d_sink);
    run<8>("v_lshl_add_u32", d_cyc, d_sink);
    run<9>("v_alignbit_b32", d_cyc, d_sink);
    run<11>("v_and_or_b32", d_cyc, d_sink);
    run<6>("v_perm_b32", d_cyc, d_sink);
    run<13>("v_add_co_u32", d_cyc, d_sink);
    run<7>("v_add_u32_sdwa BYTE_1", d_cyc, d_sink);
    run<15>("v_pk_add_u16", d_cyc, d_sink);
    run<4>("v_mul_u32_u24", d_cyc, d_sink);
    run<17>("v_mul_u32_u24_sdwa", d_cyc, d_sink);
    run<18>("v_mul_hi_u32_u24", d_cyc, d_sink);
    run<5>("v_mad_u32_u24", d_cyc, d_sink);
    run<16>("v_dot4_u32_u8", d_cyc, d_sink);
    run<2>("v_mul_lo_u32", d_cyc, d_sink);
    run<3>("v_mul_hi_u32", d_cyc, d_sink);
    run<1>("v_mad_u64_u32 (v*s)", d_cyc, d_sink);
    run<12>("v_mad_u64_u32 (v*v)", d_cyc, d_sink);
    return 0;
}
