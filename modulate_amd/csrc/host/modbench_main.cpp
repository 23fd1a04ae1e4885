// modbench_main.cpp -- the bench.py measurement without Python: C ABI only (include/modgpu.h).
//   modbench [part_bytes=4294967296] [steps=20] [warmup=3] [device=0]
// A step is one encrypt pass + one decrypt pass over an HBM-resident part; prints payload GB/s and
// the HBM read+write GB/s of the mean launch (HIP events on the launch stream, inside the library).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../../include/modgpu.h"
#include "../../../include/modgpu_testing.h"

#define TRY( x ) do { int rc_ = ( x ); if( rc_ != MODGPU_OK ) { std::printf( "modgpu error %d: %s\n", rc_, modgpu_last_error() ); return 1; } } while( 0 )

int main( int argc, char** argv )
{
    const uint64_t n = argc > 1 ? std::strtoull( argv[ 1 ], nullptr, 0 ) : ( 1ull << 32 );
    const int steps = argc > 2 ? std::atoi( argv[ 2 ] ) : 20;
    const int warmup = argc > 3 ? std::atoi( argv[ 3 ] ) : 3;
    const int device = argc > 4 ? std::atoi( argv[ 4 ] ) : 0;
    if( modgpu_device_count() < 1 ) { std::printf( "no HIP device\n" ); return 1; }
    void* part = nullptr;
    TRY( modgpu_alloc( &part, n, device ) );
    std::vector< unsigned char > tile( 64u << 20 );
    unsigned int x = 12345;
    for( auto& b : tile ) { x = x * 1664525u + 1013904223u; b = (unsigned char)( x >> 24 ); }
    for( uint64_t off = 0; off < n; off += tile.size() )
        TRY( modgpu_h2d( (char*)part + off, tile.data(), std::min< uint64_t >( tile.size(), n - off ), device ) );
    const int32_t key = (int32_t)MODGPU_KEY_PS4;
    float ms = 0.f;
    if( warmup > 0 ) TRY( modgpu_time_cycle_device( part, n, key, 0, device, nullptr, 2 * warmup, &ms ) );
    TRY( modgpu_sync( device, nullptr ) );
    auto t0 = std::chrono::steady_clock::now();
    TRY( modgpu_time_cycle_device( part, n, key, 0, device, nullptr, 2 * steps, &ms ) );
    TRY( modgpu_sync( device, nullptr ) );
    double dt = std::chrono::duration< double >( std::chrono::steady_clock::now() - t0 ).count();
    std::vector< unsigned char > back( 1u << 20 );
    TRY( modgpu_d2h( back.data(), part, std::min< uint64_t >( back.size(), n ), device ) );
    bool restored = std::equal( back.begin(), back.begin() + (long)std::min< uint64_t >( back.size(), n ), tile.begin() );
    std::printf( "{\"part_bytes\": %llu, \"steps\": %d, \"payload_GBps\": %.1f, \"ms_per_launch\": %.4f, \"hbm_read_write_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"even_passes_restore_input\": %s}\n",
                 (unsigned long long)n, steps, 2.0 * steps * (double)n / dt / 1e9, ms, 2.0 * (double)n / ( ms * 1e-3 ) / 1e9,
                 2.0 * (double)n / ( ms * 1e-3 ) / 8e12, restored ? "true" : "false" );
    TRY( modgpu_free( part, device ) );
    return restored ? 0 : 2;
}
