// modgpu_stub_for_sanitizer.cpp -- NOT part of the product.  Stands in for libmodgpu.so in the
// `make sanitize` build only, so that the host mirror's own logic (header serialise / parse,
// part split, file I/O) can run under ASan/UBSan on a machine without a GPU.  Every cipher entry
// point reports "no device", exactly as the real library does there: nothing is computed here.
#include "../../../include/modgpu.h"

extern "C" {
const char* modgpu_last_error( void ) { return "sanitizer stub: no HIP device"; }
int modgpu_cycle_host( uint8_t*, uint64_t, int32_t, uint64_t, int ) { return MODGPU_ERR_NO_DEVICE; }
int modgpu_cycle_parts_host( uint8_t* const*, const uint64_t*, int, int32_t, int ) { return MODGPU_ERR_NO_DEVICE; }
int modgpu_device_count( void ) { return 0; }
int modgpu_cycle_file_to_host( const char*, uint64_t, uint8_t*, uint64_t, int32_t, uint64_t, int ) { return MODGPU_ERR_NO_DEVICE; }
int modgpu_cycle_host_to_file( const uint8_t*, uint64_t, const char*, int32_t, uint64_t, int ) { return MODGPU_ERR_NO_DEVICE; }
}
