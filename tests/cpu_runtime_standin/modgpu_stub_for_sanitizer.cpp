// modgpu_stub_for_sanitizer.cpp -- NOT part of the product.  Stands in for libmodgpu.so's HIP side in
// the `make sanitize` build only, so that the host mirror's own logic (header serialise / parse,
// part split, file I/O) can run under ASan/UBSan on a machine without a GPU.  Every GPU entry point
// reports "no device", exactly as the real library does there; the host loop is the product's own
// scalar_path.cpp, compiled into the same build, so it runs under the sanitizers too.
#include <cstdlib>

#include "../../include/modgpu.h"
#include "../../modulate_amd/csrc/scalar_path.h"

extern "C" {
const char* modgpu_last_error( void ) { return "sanitizer stub: no HIP device"; }
int modgpu_cycle_host( uint8_t*, uint64_t, int32_t, uint64_t, int ) { return MODGPU_ERR_NO_DEVICE; }
int modgpu_cycle_auto_host( uint8_t* buf, uint64_t n, int32_t key, uint64_t off, int )
{
    const char* e = std::getenv( "MODGPU_REQUIRE_GPU" );
    if( e && *e && *e != '0' ) return MODGPU_ERR_NO_DEVICE;
    modgpu_scalar_cycle( buf, n, key, off );
    return MODGPU_OK;
}
int modgpu_cycle_parts_host( uint8_t* const*, const uint64_t*, int, int32_t, int ) { return MODGPU_ERR_NO_DEVICE; }
int modgpu_device_count( void ) { return 0; }
int modgpu_gpu_required( void )
{
    const char* e = std::getenv( "MODGPU_REQUIRE_GPU" );
    return e && *e && *e != '0';
}
int modgpu_cycle_file_to_host( const char*, uint64_t, uint8_t*, uint64_t, int32_t, uint64_t, int ) { return MODGPU_ERR_NO_DEVICE; }
int modgpu_cycle_host_to_file( const uint8_t*, uint64_t, const char*, int32_t, uint64_t, int ) { return MODGPU_ERR_NO_DEVICE; }
int modgpu_host_alloc( void** p, uint64_t n ) { *p = std::malloc( n ? n : 1 ); return *p ? MODGPU_OK : MODGPU_ERR_INVALID; }
int modgpu_host_alloc_parts( void** p, const uint64_t* sizes, int n, int )
{
    uint64_t total = 0;
    for( int i = 0; i < n; ++i ) total += sizes[ i ];
    return modgpu_host_alloc( p, total );
}
int modgpu_host_free( void* p ) { std::free( p ); return MODGPU_OK; }
int modgpu_host_is_pinned( const void*, uint64_t ) { return 0; }
}
