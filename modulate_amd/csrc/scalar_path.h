// scalar_path.h -- the library's own host loop (scalar_path.cpp); no HIP, no oracle.
#pragma once
#include <cstdint>

// buf[j] ^= ks[stream_off + j] for j < n, in place, on the calling host (threads for >= 8 MiB).
void modgpu_scalar_cycle(uint8_t *buf, uint64_t n, int32_t key, uint64_t stream_off);
