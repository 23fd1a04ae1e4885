// scalar_path.h -- the library's own host loop (scalar_path.cpp).  No HIP.
#pragma once
#include <cstdint>

enum : int { MODGPU_ISA_AUTO = -1, MODGPU_ISA_GENERIC = 0, MODGPU_ISA_AVX2 = 1, MODGPU_ISA_AVX512 = 2, MODGPU_ISA_COUNT = 3 };

// buf[j] ^= ks[stream_off + j].  isa: MODGPU_ISA_AUTO = the best body this CPU runs (latched once per process;
// MODGPU_HOST_ISA overrides), or one body by name -- an unusable one falls back to the automatic choice.
void modgpu_scalar_cycle(uint8_t *buf, uint64_t n, int32_t key, uint64_t stream_off, int isa = MODGPU_ISA_AUTO);
int modgpu_scalar_isa();                       // the automatic choice
const char *modgpu_scalar_isa_name(int isa);   // "generic" / "avx2" / "avx512"
bool modgpu_scalar_isa_usable(int isa);
// threads (the caller included) a call over n bytes spreads its spans over: MODGPU_HOST_THREADS (default min(cores, 32)),
// one per 2 MiB, never more than the control group's CPU quota or the caller's affinity mask allow
unsigned modgpu_scalar_threads_for(uint64_t n);
unsigned long long modgpu_scalar_pool_threads(); // worker threads started so far (they park between calls)
// out[0] = MODGPU_HOST_THREADS as latched, out[1] = the control group's CPU limit (0: none known), out[2] = CPUs in the calling
// thread's affinity mask, out[3] = worker threads started so far
void modgpu_scalar_info(uint64_t out[4]);
