// modbench_main.cpp -- measurements without Python: C ABI only (include/modgpu.h).
//
//   modbench [part_bytes=4294967296] [steps=20] [warmup=3] [device=0]
//       bench.py's measurement: a step is one encrypt pass + one decrypt pass over an HBM-resident part; prints payload
//       GB/s and the HBM read+write GB/s of the mean launch (HIP events on the launch stream, inside the library).
//   modbench --parts N [--devices d0,d1,... | a..b] [--part-bytes B] [--steps S] [--warmup W]
//       ONE process driving N resident parts, part i on devices[i mod len] (BASELINE config 3's shape), through
//       modgpu_cycle_parts_device: aggregate GB/s of the whole job by the wall clock, pass 1 checked against the
//       library's own closed form on windows, an even number of passes against the input.
//   modbench --hostcall
//       modgpu_cycle_host (kernel) against modgpu_cycle_scalar_host (host loop) per call from 64 B to 2 MiB, and the
//       crossover MODGPU_MIN_GPU_BYTES should sit at; per-core and all-core GB/s of every host-loop body.
//   modbench --hostcall --trace
//       the staged route's host-side timeline (modgpu_host_trace) for ONE call over a pageable buffer of 16 ... 256 MiB:
//       when the slots were there, when each pipeline started, and per pipeline what its time went into -- filling slots
//       (memcpy in), waiting for kernels, draining slots (memcpy out); full event list for the 64 MiB call.  Then two
//       callers at once on one GPU (64 MiB each) against one caller alone.
//   modbench --route pinned|staged|file_pageable|file_pinned --mib N [--reps R] [--socket near|far] [--dir D]
//       R calls of modgpu_cycle_host over ONE buffer of N MiB -- page-locked (modgpu_host_alloc: cycled in place by one kernel
//       across PCIe) or pageable (the staged route) -- with the host-side timeline on for the whole run.  Prints every call's wall
//       time and, one line each, every kernel launch the library made (thread id, call, pipeline, piece, bytes): run under
//       `rocprofv3 --kernel-trace`, tools/summarize_pcie_trace.py joins those lines with the trace's dispatches by thread id and
//       order and gives bytes / duration per kernel, the share of the wall clock with a kernel running, and the gaps on each lane.
//   modbench --alloc
//       what the part buffer costs: modgpu_host_alloc against modgpu_host_alloc_parts, and the kernel's rate on each.
//   modbench --numa
//       the kernel across PCIe on page-locked memory of each NUMA node (the GPU's own and the others).
//   modbench --files DIR [--bytes B]...
//       the file routes (modgpu_cycle_file / _file_to_host / _host_to_file) beside their two ceilings: the same
//       pread / pwrite schedule with the cipher skipped (I/O only) and the cipher with the I/O skipped (GPU only).
#include <fcntl.h>
#include <sys/stat.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/modgpu.h"
#include "../../../include/modgpu_testing.h"

#define TRY( x ) do { int rc_ = ( x ); if( rc_ != MODGPU_OK ) { std::printf( "modgpu error %d: %s\n", rc_, modgpu_last_error() ); return 1; } } while( 0 )

namespace
{
const int32_t kKey = (int32_t)MODGPU_KEY_PS4;
double Now() { return std::chrono::duration< double >( std::chrono::steady_clock::now().time_since_epoch() ).count(); }

std::vector< unsigned char > Tile( size_t n, unsigned int seed )
{
    std::vector< unsigned char > t( n );
    unsigned int x = seed;
    for( auto& b : t ) { x = x * 1664525u + 1013904223u; b = (unsigned char)( x >> 24 ); }
    return t;
}

int Classic( uint64_t n, int steps, int warmup, int device )
{
    void* part = nullptr;
    TRY( modgpu_alloc( &part, n, device ) );
    const std::vector< unsigned char > tile = Tile( 64u << 20, 12345 );
    for( uint64_t off = 0; off < n; off += tile.size() )
        TRY( modgpu_h2d( (char*)part + off, tile.data(), std::min< uint64_t >( tile.size(), n - off ), device ) );
    float ms = 0.f;
    if( warmup > 0 ) TRY( modgpu_time_cycle_device( part, n, kKey, 0, device, nullptr, 2 * warmup, &ms ) );
    TRY( modgpu_sync( device, nullptr ) );
    const double t0 = Now();
    TRY( modgpu_time_cycle_device( part, n, kKey, 0, device, nullptr, 2 * steps, &ms ) );
    TRY( modgpu_sync( device, nullptr ) );
    const double dt = Now() - t0;
    std::vector< unsigned char > back( 1u << 20 );
    TRY( modgpu_d2h( back.data(), part, std::min< uint64_t >( back.size(), n ), device ) );
    const bool restored = std::equal( back.begin(), back.begin() + (long)std::min< uint64_t >( back.size(), n ), tile.begin() );
    std::printf( "{\"part_bytes\": %llu, \"steps\": %d, \"payload_GBps\": %.1f, \"ms_per_launch\": %.4f, \"hbm_read_write_GBps\": %.1f, \"frac_of_8TBps\": %.4f, \"even_passes_restore_input\": %s}\n",
                 (unsigned long long)n, steps, 2.0 * steps * (double)n / dt / 1e9, ms, 2.0 * (double)n / ( ms * 1e-3 ) / 1e9,
                 2.0 * (double)n / ( ms * 1e-3 ) / 8e12, restored ? "true" : "false" );
    TRY( modgpu_free( part, device ) );
    return restored ? 0 : 2;
}

// ---- --parts: one process, N resident parts, N devices --------------------------------------------------------------
int Parts( int nParts, std::vector< int > devices, uint64_t n, int steps, int warmup )
{
    const int avail = modgpu_device_count();
    if( devices.empty() ) for( int d = 0; d < std::min( nParts, avail ); ++d ) devices.push_back( d );
    std::vector< void* > parts( (size_t)nParts, nullptr );
    std::vector< uint64_t > sizes( (size_t)nParts, n );
    std::vector< int > where( (size_t)nParts );
    const std::vector< unsigned char > tile = Tile( 64u << 20, 777 );
    for( int i = 0; i < nParts; ++i )
    {
        where[ i ] = devices[ (size_t)i % devices.size() ];
        TRY( modgpu_alloc( &parts[ i ], n, where[ i ] ) );
        for( uint64_t off = 0; off < n; off += tile.size() )
            TRY( modgpu_h2d( (char*)parts[ i ] + off, tile.data(), std::min< uint64_t >( tile.size(), n - off ), where[ i ] ) );
    }
    auto pass = [ & ]() { return modgpu_cycle_parts_device( parts.data(), sizes.data(), where.data(), nParts, kKey ); };
    // pass 1 of the warm-up doubles as the check: ciphertext ^ plaintext == low8(state) ^ 0xFF on windows of every part
    TRY( pass() );
    bool ok = true;
    std::vector< unsigned char > win( 4096 );
    for( int i = 0; i < nParts && ok; ++i )
        for( uint64_t off : { (uint64_t)0, n / 2 + 13, n - std::min< uint64_t >( n, win.size() ) } )
        {
            const uint64_t len = std::min< uint64_t >( win.size(), n - off );
            TRY( modgpu_d2h( win.data(), (char*)parts[ i ] + off, len, where[ i ] ) );
            for( uint64_t j = 0; j < len; j += 97 )
                ok = ok && (unsigned char)( win[ j ] ^ tile[ ( off + j ) % tile.size() ] ) == (unsigned char)( ~modgpu_state_at( kKey, off + j ) & 0xFF );
        }
    TRY( pass() );
    for( int w = 1; w < warmup; ++w ) { TRY( pass() ); TRY( pass() ); }
    modgpu_path_stats_t st0{}, st1{};
    modgpu_path_stats( &st0, 0 );
    const double t0 = Now();
    for( int s = 0; s < steps; ++s ) { TRY( pass() ); TRY( pass() ); }
    const double dt = Now() - t0;
    modgpu_path_stats( &st1, 0 );
    modgpu_launch_info_t last{};
    modgpu_last_launch( &last );
    for( int i = 0; i < nParts && ok; ++i )
    {
        TRY( modgpu_d2h( win.data(), (char*)parts[ i ] + ( n > win.size() ? n - win.size() : 0 ), std::min< uint64_t >( win.size(), n ), where[ i ] ) );
        const uint64_t off = n > win.size() ? n - win.size() : 0;
        for( uint64_t j = 0; j < std::min< uint64_t >( win.size(), n ); ++j ) ok = ok && win[ j ] == tile[ ( off + j ) % tile.size() ];
    }
    // Each device's parts alone, and ONE part alone (the N = 1 line): what the aggregate is to be held against.  Fewer steps:
    // these are reference points, the aggregate above is the measurement.
    std::vector< int > distinct;
    for( int d : where ) if( std::find( distinct.begin(), distinct.end(), d ) == distinct.end() ) distinct.push_back( d );
    const int refSteps = std::max( 2, steps / 4 );
    auto timeSubset = [ & ]( const std::vector< int >& idx ) -> double {
        std::vector< void* > p; std::vector< uint64_t > z; std::vector< int > w;
        for( int i : idx ) { p.push_back( parts[ (size_t)i ] ); z.push_back( n ); w.push_back( where[ (size_t)i ] ); }
        auto go = [ & ]() { return modgpu_cycle_parts_device( p.data(), z.data(), w.data(), (int)p.size(), kKey ); };
        if( go() != MODGPU_OK || go() != MODGPU_OK ) return -1.0;
        const double t = Now();
        for( int s2 = 0; s2 < refSteps; ++s2 ) { if( go() != MODGPU_OK || go() != MODGPU_OK ) return -1.0; }
        return 2.0 * refSteps * idx.size() * (double)n / ( Now() - t ) / 1e9;
    };
    std::string perDevice;
    for( size_t k = 0; k < distinct.size(); ++k )
    {
        std::vector< int > idx;
        for( int i = 0; i < nParts; ++i ) if( where[ (size_t)i ] == distinct[ k ] ) idx.push_back( i );
        char b[ 64 ];
        std::snprintf( b, sizeof b, "%s{\"device\": %d, \"parts\": %zu, \"GBps\": %.1f}", k ? ", " : "", distinct[ k ], idx.size(), timeSubset( idx ) );
        perDevice += b;
    }
    const double n1 = timeSubset( { 0 } );
    std::string devs;
    for( size_t i = 0; i < devices.size(); ++i ) devs += ( i ? "," : "" ) + std::to_string( devices[ i ] );
    std::printf( "{\"mode\": \"parts\", \"parts\": %d, \"devices\": [%s], \"logical_devices_visible\": %d, \"part_bytes\": %llu, \"steps\": %d, "
                 "\"aggregate_payload_GBps\": %.1f, \"aggregate_hbm_read_write_GBps\": %.1f, \"ms_per_pass_over_all_parts\": %.4f, "
                 "\"kernel_launches_per_pass\": %.2f, \"last_kernel\": \"%s\", \"bit_exact_windows\": %s}\n",
                 nParts, devs.c_str(), avail, (unsigned long long)n, steps, 2.0 * steps * nParts * (double)n / dt / 1e9,
                 4.0 * steps * nParts * (double)n / dt / 1e9, dt / ( 2.0 * steps ) * 1e3,
                 (double)( st1.gpu_launches - st0.gpu_launches ) / ( 2.0 * steps ), last.kernel ? last.kernel : "?", ok ? "true" : "false" );
    // The same job in the bench contract's shape (bench.py's line at --gpus N comes from N processes; this is ONE process
    // driving N devices), with what bench.py never reports itself: each device's own rate and the efficiency against N x the
    // one-part line.  On aliased devices (MODGPU_DEVICE_ALIAS) every "device" is the same GPU and the efficiency says so.
    const double agg = 2.0 * steps * nParts * (double)n / dt / 1e9;
    std::printf( "{\"metric\": \"GB/s encrypt+decrypt over synthetic .ark parts; %% HBM peak at 1/2/4/8 GPU\", \"value\": %.2f, \"unit\": \"GB/s\", \"n_gpus\": %zu, "
                 "\"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", \"vs_baseline\": null, \"dtype\": \"u8\", "
                 "\"data\": \"synthetic\", \"config\": {\"workload\": \"config 3 in ONE process: %d x %llu B resident parts over %zu device(s), modgpu_cycle_parts_device\", "
                 "\"parallelism\": \"parts%zu\"}, \"per_device\": [%s], \"n1_value\": %.2f, \"efficiency_vs_n_times_n1\": %.4f, \"bit_exact_windows\": %s}\n",
                 agg, distinct.size(), steps, warmup, dt / steps * 1e3, nParts, (unsigned long long)n, distinct.size(), distinct.size(), perDevice.c_str(), n1,
                 n1 > 0 ? agg / ( n1 * (double)distinct.size() ) : 0.0, ok ? "true" : "false" );
    for( int i = 0; i < nParts; ++i ) TRY( modgpu_free( parts[ i ], where[ i ] ) );
    return ok ? 0 : 2;
}

// ---- --hostcall: kernel vs host loop per call ------------------------------------------------------------------------
template < typename F > double MedianUs( F&& call, int reps )
{
    std::vector< double > t( (size_t)reps );
    for( int i = 0; i < reps; ++i )
    {
        const double t0 = Now();
        call();
        t[ i ] = ( Now() - t0 ) * 1e6;
    }
    std::sort( t.begin(), t.end() );
    return t[ t.size() / 2 ];
}

int HostCall()
{
    const bool gpu = modgpu_device_count() > 0;
    std::printf( "== per-call latency, caller-owned pageable host buffer (what CEncryptionCycler::Cycle gets): median of N calls\n" );
    std::printf( "   host loop body: %s   |   MODGPU_MIN_GPU_BYTES as latched: %llu   |   GPU %s\n", modgpu_host_loop_isa(),
                 (unsigned long long)modgpu_min_gpu_bytes(), gpu ? "present" : "ABSENT (host-loop column only)" );
    {
        uint64_t li[ 4 ];
        modgpu_host_loop_info( li );
        std::printf( "   host loop threads: at most %llu per call (MODGPU_HOST_THREADS), control-group CPU limit %llu (0 = none), %llu CPUs in this thread's affinity mask; policy above the threshold: %s\n",
                     (unsigned long long)li[ 0 ], (unsigned long long)li[ 1 ], (unsigned long long)li[ 2 ], modgpu_host_policy() );
    }
    std::printf( "   %10s  %14s  %14s  %10s\n", "bytes", "kernel (us)", "host loop (us)", "faster" );
    uint64_t crossover = 0;
    bool crossed = false;
    for( uint64_t n : { 64ull, 256ull, 1024ull, 4092ull, 8192ull, 16384ull, 32768ull, 49152ull, 65536ull, 98304ull, 131072ull, 196608ull, 262144ull, 393216ull, 524288ull, 1048576ull, 2097152ull } )
    {
        std::vector< unsigned char > buf = Tile( n, (unsigned int)n );
        const int reps = n <= 65536 ? 2000 : 400;
        double g = -1;
        if( gpu )
        {
            for( int i = 0; i < 20; ++i ) TRY( modgpu_cycle_host( buf.data(), n, kKey, 0, 0 ) );
            g = MedianUs( [ & ] { (void)modgpu_cycle_host( buf.data(), n, kKey, 0, 0 ); }, reps );
        }
        for( int i = 0; i < 20; ++i ) TRY( modgpu_cycle_scalar_host( buf.data(), n, kKey, 0 ) );
        const double h = MedianUs( [ & ] { (void)modgpu_cycle_scalar_host( buf.data(), n, kKey, 0 ); }, reps );
        if( gpu && !crossed && g < h ) { crossover = n; crossed = true; }
        if( gpu && g >= h ) crossed = false; // keep the LAST size from which the kernel stays ahead
        std::printf( "   %10llu  %14.1f  %14.1f  %10s\n", (unsigned long long)n, g, h, !gpu ? "-" : ( g < h ? "kernel" : "host loop" ) );
    }
    if( gpu && crossed ) std::printf( "   kernel ahead from %llu bytes up\n", (unsigned long long)crossover );
    if( gpu && !crossed ) std::printf( "   the host loop is ahead at every size of this table\n" );
    // part-sized buffers: the kernel route (staged through pinned slots, 8 pipelines) against the host loop on ONE thread
    // (2 MiB pieces, each below the size from which the loop spreads over threads) and on its threads
    std::printf( "== part-sized pageable buffers: GB/s of payload; kernel route best of 5-10 calls (all sizes first), host loop best of 3\n" );
    std::printf( "   %10s  %14s  %18s  %18s\n", "MiB", "kernel route", "host loop 1 thread", "host loop threads" );
    // (the kernel route of every size first, the host loop afterwards: sixteen host-loop threads that have just used up the
    //  control group's CPU quota leave the staging pipelines' copy threads throttled for the next measurement -- round 5 saw the
    //  kernel column 25-40 % low when the two were interleaved per size)
    const uint64_t partSizes[] = { 4ull << 20, 8ull << 20, 16ull << 20, 32ull << 20, 64ull << 20, 256ull << 20, 1ull << 30 };
    auto best = []( int reps, auto&& run ) { double b = 1e30; for( int i = 0; i < reps; ++i ) { const double t0 = Now(); run(); b = std::min( b, Now() - t0 ); } return b; };
    std::vector< double > kernelSeconds;
    for( uint64_t n : partSizes )
    {
        std::vector< unsigned char > buf( n, 0x3C );
        double g = -1;
        if( gpu )
        {
            for( int i = 0; i < 2; ++i ) TRY( modgpu_cycle_host( buf.data(), n, kKey, 0, 0 ) );
            g = best( n <= ( 64ull << 20 ) ? 10 : 5, [ & ] { (void)modgpu_cycle_host( buf.data(), n, kKey, 0, 0 ); } );
        }
        kernelSeconds.push_back( g );
    }
    size_t row = 0;
    for( uint64_t n : partSizes )
    {
        std::vector< unsigned char > buf( n, 0x3C );
        const double g = kernelSeconds[ row++ ];
        const double h1 = best( 3, [ & ] { for( uint64_t off = 0; off < n; off += 2ull << 20 ) (void)modgpu_cycle_scalar_host( buf.data() + off, std::min< uint64_t >( 2ull << 20, n - off ), kKey, off ); } );
        const double ht = best( 3, [ & ] { (void)modgpu_cycle_scalar_host( buf.data(), n, kKey, 0 ); } );
        std::printf( "   %10llu  %14.1f  %18.1f  %18.1f\n", (unsigned long long)( n >> 20 ), gpu ? n / g / 1e9 : -1.0, n / h1 / 1e9, n / ht / 1e9 );
    }
    std::printf( "== host loop throughput by body (in place, warm; one thread below 4 MiB, MODGPU_HOST_THREADS / all cores above)\n" );
    for( const char* isa : { "generic", "avx2", "avx512" } )
    {
        std::vector< unsigned char > probe( 64 );
        if( modgpu_cycle_scalar_host_isa( probe.data(), probe.size(), kKey, 0, isa ) != MODGPU_OK ) { std::printf( "   %-8s not available on this CPU\n", isa ); continue; }
        for( uint64_t n : { 1ull << 20, 3ull << 20, 256ull << 20, 1ull << 30 } )
        {
            std::vector< unsigned char > buf( n, 0x5A );
            (void)modgpu_cycle_scalar_host_isa( buf.data(), n, kKey, 0, isa );
            const int reps = n <= ( 3ull << 20 ) ? 50 : 3;
            const double us = MedianUs( [ & ] { (void)modgpu_cycle_scalar_host_isa( buf.data(), n, kKey, 0, isa ); }, reps );
            std::printf( "   %-8s %6llu MiB  %9.1f us  %7.2f GB/s  (%s)\n", isa, (unsigned long long)( n >> 20 ), us, n / us / 1e3,
                         n < ( 4ull << 20 ) ? "1 thread" : "threads" );
        }
    }
    return 0;
}

// ---- --hostcall --trace: where one staged call's time goes -----------------------------------------------
int HostTrace()
{
    if( modgpu_device_count() < 1 ) { std::printf( "no HIP device\n" ); return 1; }
    static const char* kKind[] = { "call_begin", "slots", "posted", "pipe_start", "fill_begin", "fill_end", "launched", "sync_begin", "sync_end", "drain_end", "pipe_end", "call_end", "failed", "rescued", "ready" };
    uint64_t tun[ 4 ], chk[ 4 ];
    modgpu_host_tunables( tun );
    modgpu_host_chunking( chk );
    std::printf( "== staged route (pageable caller memory -> pinned slot -> kernel across PCIe on the slot -> back), one call traced per size\n" );
    std::printf( "   pipelines <= %llu, slot <= %llu MiB, a buffer is cut into ~%llu chunks of >= %llu MiB, each pipeline's first and last chunk %llu KiB, kernels queued on %llu lane(s)\n", (unsigned long long)tun[ 0 ],
                 (unsigned long long)( tun[ 1 ] >> 20 ), (unsigned long long)chk[ 0 ], (unsigned long long)( chk[ 1 ] >> 20 ), (unsigned long long)( chk[ 2 ] >> 10 ), (unsigned long long)chk[ 3 ] );
    for( uint64_t mib : { 16ull, 32ull, 64ull, 128ull, 256ull } )
    {
        const uint64_t n = mib << 20;
        std::vector< unsigned char > buf( n, 0x3C );
        double best = 1e30;
        for( int i = 0; i < 6; ++i ) { const double t0 = Now(); TRY( modgpu_cycle_host( buf.data(), n, kKey, 0, 0 ) ); best = std::min( best, Now() - t0 ); }
        modgpu_host_trace( 1 );
        const double t0 = Now();
        TRY( modgpu_cycle_host( buf.data(), n, kKey, 0, 0 ) );
        const double traced = Now() - t0;
        modgpu_host_trace( 0 );
        std::vector< modgpu_host_trace_event_t > ev( (size_t)modgpu_host_trace_read( nullptr, 0 ) );
        modgpu_host_trace_read( ev.data(), (int)ev.size() );
        if( ev.empty() ) continue;
        const uint64_t z = ev.front().t_ns;
        auto us = [ & ]( uint64_t t ) { return ( t - z ) * 1e-3; };
        struct Pipe { double start = -1, end = 0, fill = 0, sync = 0, drain = 0, launch = 0, first_launch = -1; uint64_t chunks = 0; double t_fill = 0, t_sync = 0, t_launch = 0; };
        std::vector< Pipe > pipes( 32 );
        double slots_at = 0, posted_at = 0, end_at = 0, slot_bytes = 0;
        uint64_t n_pipes = 0;
        for( const auto& e : ev )
        {
            const double t = us( e.t_ns );
            if( e.kind == MODGPU_TRACE_SLOTS ) { slots_at = t; n_pipes = e.chunk; slot_bytes = (double)e.bytes; }
            if( e.kind == MODGPU_TRACE_POSTED ) posted_at = t;
            if( e.kind == MODGPU_TRACE_CALL_END ) end_at = t;
            if( e.pipe < 0 || e.pipe >= 32 ) continue;
            Pipe& p = pipes[ (size_t)e.pipe ];
            switch( e.kind )
            {
            case MODGPU_TRACE_PIPE_START: p.start = t; break;
            case MODGPU_TRACE_FILL_BEGIN: p.t_fill = t; break;
            case MODGPU_TRACE_FILL_END: p.fill += t - p.t_fill; p.t_launch = t; ++p.chunks; break;
            case MODGPU_TRACE_READY: // (host-fed call: marking the chunk ready is this route's "launch")
            case MODGPU_TRACE_LAUNCHED: p.launch += t - p.t_launch; if( p.first_launch < 0 ) p.first_launch = t; break;
            case MODGPU_TRACE_SYNC_BEGIN: p.t_sync = t; break;
            case MODGPU_TRACE_SYNC_END: p.sync += t - p.t_sync; p.t_fill = t; break;
            case MODGPU_TRACE_DRAIN_END: p.drain += t - p.t_fill; break;
            case MODGPU_TRACE_PIPE_END: p.end = t; break;
            default: break;
            }
        }
        std::printf( "-- %llu MiB: best of 6 %.3f ms = %.1f GB/s; the traced call %.3f ms; %llu pipelines, slots of %.1f MiB; slots there at %.0f us, pipelines posted at %.0f us, call ends at %.0f us\n",
                     (unsigned long long)mib, best * 1e3, n / best / 1e9, traced * 1e3, (unsigned long long)n_pipes, slot_bytes / ( 1 << 20 ), slots_at, posted_at, end_at );
        std::printf( "   pipe  start_us  first_launch_us  chunks  fill_us  launch_us  wait_for_kernel_us  drain_us   end_us  (fill + launch + wait + drain = busy)\n" );
        double latest_start = 0, sum_fill = 0, sum_sync = 0, sum_drain = 0, sum_launch = 0;
        for( size_t k = 0; k < pipes.size(); ++k )
        {
            const Pipe& p = pipes[ k ];
            if( p.start < 0 ) continue;
            latest_start = std::max( latest_start, p.start );
            sum_fill += p.fill; sum_sync += p.sync; sum_drain += p.drain; sum_launch += p.launch;
            std::printf( "   %4zu  %8.0f  %15.0f  %6llu  %7.0f  %9.0f  %18.0f  %8.0f  %7.0f\n", k, p.start, p.first_launch, (unsigned long long)p.chunks, p.fill, p.launch, p.sync, p.drain, p.end );
        }
        const double np_ = (double)std::max< uint64_t >( n_pipes, 1 );
        std::printf( "   per pipeline on average: fill %.0f us, launch calls %.0f us, waiting for kernels %.0f us, drain %.0f us; last pipeline started at %.0f us;"
                     " the link alone (50 GB/s each way) needs %.0f us\n", sum_fill / np_, sum_launch / np_, sum_sync / np_, sum_drain / np_, latest_start, n / 50e9 * 1e6 );
        if( mib == 64 )
        {
            std::printf( "   every event of the 64 MiB call: t_us kind pipe chunk bytes\n" );
            for( const auto& e : ev ) std::printf( "     %8.1f %-11s %3d %4llu %9llu\n", us( e.t_ns ), kKind[ e.kind ], e.pipe, (unsigned long long)e.chunk, (unsigned long long)e.bytes );
        }
    }
    // two callers on ONE GPU at once (round 3 serialised them end to end on the device's staging mutex)
    std::printf( "== concurrent callers on one GPU, pageable buffers, GB/s of payload summed over the callers (best of 5 rounds)\n" );
    for( uint64_t mib : { 16ull, 64ull, 256ull } )
    {
        const uint64_t n = mib << 20;
        for( int callers : { 1, 2, 4 } )
        {
            std::vector< std::vector< unsigned char > > bufs( (size_t)callers, std::vector< unsigned char >( n, 0x11 ) );
            double best = 1e30;
            for( int r = 0; r < 6; ++r )
            {
                const double t0 = Now();
                std::vector< std::thread > ts;
                for( int c = 1; c < callers; ++c ) ts.emplace_back( [ &, c ] { (void)modgpu_cycle_host( bufs[ (size_t)c ].data(), n, kKey, 0, 0 ); } );
                (void)modgpu_cycle_host( bufs[ 0 ].data(), n, kKey, 0, 0 );
                for( auto& t : ts ) t.join();
                if( r > 0 ) best = std::min( best, Now() - t0 );
            }
            std::printf( "   %4llu MiB x %d caller%s: %7.3f ms  %6.1f GB/s\n", (unsigned long long)mib, callers, callers > 1 ? "s" : " ", best * 1e3, callers * (double)n / best / 1e9 );
        }
    }
    uint64_t ps[ 6 ];
    modgpu_host_pool_stats( ps );
    std::printf( "   staging pool: %llu worker threads started in this process, %llu pipelines run by them, %llu waits for a slot, %llu calls began while another was in flight\n",
                 (unsigned long long)ps[ 0 ], (unsigned long long)ps[ 1 ], (unsigned long long)ps[ 2 ], (unsigned long long)ps[ 3 ] );
    return 0;
}

// ---- --route: one host-buffer route, R calls, every launch listed (to be joined with a rocprofv3 kernel trace) -------------------
std::string gRouteDir = "/dev/shm"; // --dir D: where the file routes' part file is put
uint64_t gRouteOffset = 4; // --offset K: the buffer starts K bytes behind a page boundary (the reference's callers pass buf + 4)
std::string gRouteSocket;  // --socket near|far: this thread (and so the pages it touches) on the GPU's NUMA node / on another one, before anything
                           // is allocated -- a profile that does not depend on where the scheduler happened to start the process
void BindToSocket( const std::string& which )
{
    const int gpuNode = modgpu_device_numa_node( 0 );
    if( gpuNode < 0 ) { std::printf( "--socket %s: the GPU's NUMA node is unknown, not bound\n", which.c_str() ); return; }
    for( int node = 0; node < 64; ++node )
    {
        if( ( which == "near" ) != ( node == gpuNode ) ) continue;
        std::FILE* f = std::fopen( ( "/sys/devices/system/node/node" + std::to_string( node ) + "/cpulist" ).c_str(), "r" );
        if( !f ) continue;
        char text[ 4096 ] = {};
        const size_t got = std::fread( text, 1, sizeof text - 1, f );
        std::fclose( f );
        cpu_set_t now, want;
        CPU_ZERO( &want );
        if( !got || sched_getaffinity( 0, sizeof now, &now ) != 0 ) continue;
        int count = 0;
        for( char* p = text; *p && *p != '\n'; )
        {
            const long a = std::strtol( p, &p, 10 );
            const long b = *p == '-' ? std::strtol( p + 1, &p, 10 ) : a;
            for( long c = a; c <= b && c < CPU_SETSIZE; ++c ) if( CPU_ISSET( c, &now ) ) { CPU_SET( c, &want ); ++count; }
            if( *p == ',' ) ++p;
        }
        if( count && sched_setaffinity( 0, sizeof want, &want ) == 0 ) return;
    }
    std::printf( "--socket %s: no such node with CPUs this process may use, not bound\n", which.c_str() );
}
int Route( const std::string& kind, uint64_t mib, int reps )
{
    if( modgpu_device_count() < 1 ) { std::printf( "no HIP device\n" ); return 1; }
    if( !gRouteSocket.empty() ) BindToSocket( gRouteSocket );
    const uint64_t n = mib << 20;
    // file_pageable / file_pinned: the source is a part file on tmpfs (modgpu_cycle_file_to_host: LoadArkData's part cipher), the destination this buffer
    const bool fromFile = kind == "file_pageable" || kind == "file_pinned";
    const bool pinned = kind == "pinned" || kind == "file_pinned";
    void* mem = nullptr;
    std::vector< unsigned char > pageable;
    if( pinned ) TRY( modgpu_host_alloc( &mem, n + 64 ) );
    else { pageable.assign( n + 8192, 0 ); mem = reinterpret_cast< void* >( ( reinterpret_cast< uintptr_t >( pageable.data() ) + 4095 ) & ~uintptr_t( 4095 ) ); }
    unsigned char* buf = static_cast< unsigned char* >( mem ) + ( pinned ? 4 : gRouteOffset );
    const std::vector< unsigned char > tile = Tile( 1u << 20, 777 );
    for( uint64_t off = 0; off < n; off += tile.size() ) std::memcpy( buf + off, tile.data(), std::min< uint64_t >( tile.size(), n - off ) );
    uint64_t tun[ 4 ], chk[ 4 ];
    modgpu_host_tunables( tun );
    modgpu_host_chunking( chk );
    std::printf( "route %s  bytes %llu  reps %d  pinned_as_seen_by_the_library %d  pipes<=%llu slot<=%lluMiB split~%llu chunk>=%lluMiB ramp %lluKiB lanes %llu\n", kind.c_str(),
                 (unsigned long long)n, reps, modgpu_host_is_pinned( buf, n ), (unsigned long long)tun[ 0 ], (unsigned long long)( tun[ 1 ] >> 20 ),
                 (unsigned long long)chk[ 0 ], (unsigned long long)( chk[ 1 ] >> 20 ), (unsigned long long)( chk[ 2 ] >> 10 ), (unsigned long long)chk[ 3 ] );
    { // where this process happens to live: the staged route's copies are CPU work, and which NUMA node the caller's pages are on decides their rate
        unsigned cpu = 0, node = 0;
        (void)::syscall( SYS_getcpu, &cpu, &node, nullptr );
        int pageNode = -1;
        (void)::syscall( SYS_get_mempolicy, &pageNode, nullptr, 0, buf + n / 2, 3 /* MPOL_F_NODE | MPOL_F_ADDR */ );
        std::printf( "placement: calling thread on cpu %u of NUMA node %u; the buffer's middle page on node %d; the GPU hangs off node %d\n", cpu, node, pageNode, modgpu_device_numa_node( 0 ) );
    }
    std::printf( "hashes kernel %s feed %s\n", modgpu_kernel_source_hash(), modgpu_feed_kernel_source_hash() );
    std::string path;
    if( fromFile )
    {
        path = gRouteDir + "/modbench_route_" + std::to_string( ::getpid() ) + ".part";
        std::FILE* f = std::fopen( path.c_str(), "wb" );
        if( !f || std::fwrite( buf, 1, n, f ) != n ) { std::printf( "cannot write %s\n", path.c_str() ); return 1; }
        std::fclose( f );
    }
    auto once = [ & ]() -> int { return fromFile ? modgpu_cycle_file_to_host( path.c_str(), 0, buf, n, kKey, 0, 0 ) : modgpu_cycle_host( buf, n, kKey, 0, 0 ); };
    modgpu_host_trace( 1 ); // from the first call on: the launch list below must hold EVERY launch the profiler sees
    for( int i = 0; i < 2; ++i ) TRY( once() ); // slots, workers, page faults
    std::vector< double > walls;
    for( int r = 0; r < reps; ++r )
    {
        const double t0 = Now();
        TRY( once() );
        walls.push_back( Now() - t0 );
    }
    if( fromFile )
    { // every call gave the same ciphertext; one more pass over it in memory (an untimed call of its own in the launch list) must give the file's bytes back
        ::unlink( path.c_str() );
        TRY( modgpu_cycle_host( buf, n, kKey, 0, 0 ) );
        reps = 0; // (the restore check below: "an even number of passes")
    }
    modgpu_host_trace( 0 );
    std::vector< double > sorted = walls;
    std::sort( sorted.begin(), sorted.end() );
    std::printf( "calls (after 2 untimed): best %.3f ms = %.2f GB/s, median %.3f ms = %.2f GB/s of payload (each byte crosses the link twice)\n", sorted.front() * 1e3,
                 n / sorted.front() / 1e9, sorted[ sorted.size() / 2 ] * 1e3, n / sorted[ sorted.size() / 2 ] / 1e9 );
    {
        uint64_t ps[ 6 ];
        modgpu_host_pool_stats( ps );
        std::printf( "staging: %llu of this process's calls took the slots and workers of ANOTHER NUMA node than the GPU's (the caller's pages live there)\n", (unsigned long long)ps[ 5 ] );
    }
    for( size_t r = 0; r < walls.size(); ++r ) std::printf( "call %zu wall_us %.1f\n", r + 2, walls[ r ] * 1e6 );
    std::vector< modgpu_host_trace_event_t > ev( (size_t)modgpu_host_trace_read( nullptr, 0 ) );
    modgpu_host_trace_read( ev.data(), (int)ev.size() );
    int call = -1;
    uint64_t z = ev.empty() ? 0 : ev.front().t_ns;
    for( const auto& e : ev )
    {
        if( e.kind == MODGPU_TRACE_CALL_BEGIN ) { ++call; std::printf( "callbegin %d t_us %.1f\n", call, ( e.t_ns - z ) * 1e-3 ); }
        if( e.kind == MODGPU_TRACE_CALL_END ) std::printf( "callend %d t_us %.1f\n", call, ( e.t_ns - z ) * 1e-3 );
        if( e.kind == MODGPU_TRACE_LAUNCHED ) std::printf( "launch tid %d call %d pipe %d piece %llu bytes %llu t_us %.1f\n", e.tid, call, e.pipe, (unsigned long long)e.chunk, (unsigned long long)e.bytes, ( e.t_ns - z ) * 1e-3 );
    }
    // the last call in full: what happens before its first kernel and after its last
    static const char* kKind[] = { "call_begin", "slots", "posted", "pipe_start", "fill_begin", "fill_end", "launched", "sync_begin", "sync_end", "drain_end", "pipe_end", "call_end", "failed", "rescued", "ready" };
    size_t lastBegin = 0;
    for( size_t k = 0; k < ev.size(); ++k ) if( ev[ k ].kind == MODGPU_TRACE_CALL_BEGIN ) lastBegin = k;
    for( size_t k = lastBegin; k < ev.size(); ++k )
        std::printf( "event %9.1f %-11s pipe %3d piece %4llu bytes %9llu tid %d\n", ( ev[ k ].t_ns - ev[ lastBegin ].t_ns ) * 1e-3, kKind[ ev[ k ].kind ], ev[ k ].pipe, (unsigned long long)ev[ k ].chunk, (unsigned long long)ev[ k ].bytes, ev[ k ].tid );
    bool restored = true; // an even number of passes
    if( ( reps + 2 ) % 2 == 0 )
        for( uint64_t off = 0; off < n && restored; off += tile.size() ) restored = std::memcmp( buf + off, tile.data(), std::min< uint64_t >( tile.size(), n - off ) ) == 0;
    std::printf( "even_passes_restore_input %s\n", ( reps + 2 ) % 2 ? "n/a" : ( restored ? "true" : "false" ) );
    if( pinned ) TRY( modgpu_host_free( mem ) );
    return restored ? 0 : 2;
}

// ---- --files: the file routes beside their ceilings ---------------------------------------------------------------------
// the library's chunking of a stream of n bytes (host_stream.cpp: stream_impl) at its default tunables
void Schedule( uint64_t n, uint64_t* chunk, int* pipes )
{
    uint64_t tun[ 4 ];
    modgpu_host_tunables( tun );
    const uint64_t cap = tun[ 1 ], cmin = std::min< uint64_t >( 4ull << 20, cap ), split = 16; // (a file on either side: host_stream.cpp, stream_impl)
    uint64_t c = n <= cmin ? std::max< uint64_t >( n, 1ull << 20 ) : std::min< uint64_t >( cap, std::max< uint64_t >( cmin, ( ( n / split ) + 0xFFFFF ) & ~0xFFFFFull ) );
    c = std::min( c, cap );
    const uint64_t chunks = ( n + c - 1 ) / c;
    *chunk = c;
    *pipes = (int)std::min< uint64_t >( tun[ 0 ], ( chunks + 1 ) / 2 );
}

// Writes the cache lines of [p, p + len) back and drops them from every cache: afterwards a reader finds the bytes in DRAM only --
// where bytes that the GPU wrote across PCIe are (is that why the file routes' write side is slower than the
// I/O-only harness, whose pwrite reads bytes its own pread has just put into the cache?).
__attribute__(( target( "clflushopt" ) )) void EvictFromCaches( const void* p, uint64_t len )
{
    const char* c = static_cast< const char* >( p );
    for( uint64_t o = 0; o < len; o += 64 ) _mm_clflushopt( const_cast< char* >( c + o ) );
    _mm_sfence();
}

bool ReadAll( int fd, void* p, uint64_t len, uint64_t off )
{
    for( uint64_t done = 0; done < len; )
    {
        ssize_t r = ::pread( fd, (char*)p + done, len - done, (off_t)( off + done ) );
        if( r <= 0 ) return false;
        done += (uint64_t)r;
    }
    return true;
}
bool WriteAll( int fd, const void* p, uint64_t len, uint64_t off )
{
    for( uint64_t done = 0; done < len; )
    {
        ssize_t r = ::pwrite( fd, (const char*)p + done, len - done, (off_t)( off + done ) );
        if( r <= 0 ) return false;
        done += (uint64_t)r;
    }
    return true;
}

// I/O only: `pipes` threads, thread p takes chunks p, p+pipes, ...; src/dst are a file (fd >= 0) or memory
double IoOnly( int inFd, const unsigned char* inMem, int outFd, unsigned char* outMem, uint64_t n, bool preallocate, bool coldSource = false, double* evictSeconds = nullptr )
{
    std::atomic< uint64_t > evictNs{ 0 };
    uint64_t chunk;
    int pipes;
    Schedule( n, &chunk, &pipes );
    const uint64_t chunks = ( n + chunk - 1 ) / chunk;
    const double t0 = Now();
    if( preallocate && outFd >= 0 ) (void)::posix_fallocate( outFd, 0, (off_t)n ); // inside the timed region: it is part of the job
    std::vector< std::thread > pool;
    for( int p = 0; p < pipes; ++p )
        pool.emplace_back( [ =, &chunk, &evictNs ] {
            void* slot = nullptr;
            if( modgpu_host_alloc( &slot, chunk ) != MODGPU_OK ) return;
            for( uint64_t c = (uint64_t)p; c < chunks; c += (uint64_t)pipes )
            {
                const uint64_t off = c * chunk, len = std::min( chunk, n - off );
                const void* from = inMem ? (const void*)( inMem + off ) : slot;
                if( !inMem ) ReadAll( inFd, slot, len, off );
                if( coldSource && __builtin_cpu_supports( "clflushopt" ) )
                {
                    const double e0 = Now();
                    EvictFromCaches( from, len );
                    evictNs.fetch_add( (uint64_t)( ( Now() - e0 ) * 1e9 ) );
                }
                if( outMem ) { if( from != outMem + off ) std::memcpy( outMem + off, from, len ); }
                else WriteAll( outFd, from, len, off );
            }
            modgpu_host_free( slot );
        } );
    for( auto& t : pool ) t.join();
    if( evictSeconds ) *evictSeconds = evictNs.load() * 1e-9 / std::max( pipes, 1 ); // per thread: what the eviction itself added to the wall clock
    return Now() - t0;
}

int Files( const std::string& dir, std::vector< uint64_t > sizes )
{
    if( sizes.empty() ) sizes = { 64ull << 20, 411ull * 1000 * 1000, 1ull << 32 };
    std::printf( "== file routes beside their ceilings (dir %s; GB/s of payload; best of 3)\n", dir.c_str() );
    std::printf( "   I/O only = the same chunks and threads, pread/pwrite (or memcpy) without the cipher; GPU only = modgpu_cycle_host on page-locked memory of the same size\n" );
    for( uint64_t n : sizes )
    {
        const std::string src = dir + "/modbench_src.bin", dst = dir + "/modbench_dst.bin";
        void* pinned = nullptr;
        TRY( modgpu_host_alloc( &pinned, n ) );
        unsigned char* mem = static_cast< unsigned char* >( pinned );
        {
            const std::vector< unsigned char > tile = Tile( 16u << 20, 99 );
            for( uint64_t off = 0; off < n; off += tile.size() ) std::memcpy( mem + off, tile.data(), std::min< uint64_t >( tile.size(), n - off ) );
            int fd = ::open( src.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644 );
            if( fd < 0 || !WriteAll( fd, mem, n, 0 ) ) { std::printf( "cannot write %s\n", src.c_str() ); return 1; }
            ::close( fd );
        }
        auto best = []( auto&& run ) { double b = 1e30; for( int i = 0; i < 3; ++i ) b = std::min( b, run() ); return b; };
        uint64_t chunk;
        int pipes;
        Schedule( n, &chunk, &pipes );
        std::printf( "-- %7.0f MiB  (%d pipelines x %llu MiB chunks)\n", n / 1048576.0, pipes, (unsigned long long)( chunk >> 20 ) );
        const double gpuOnly = best( [ & ] { const double t0 = Now(); (void)modgpu_cycle_host( mem, n, kKey, 0, 0 ); return Now() - t0; } );
        std::printf( "   %-44s %8.2f GB/s\n", "GPU only (pinned memory in place)", n / gpuOnly / 1e9 );
        struct Row { const char* name; double io, ioPre, route; };
        std::vector< Row > rows;
        { // file -> file
            Row r{ "modgpu_cycle_file (file -> file)", 0, 0, 0 };
            r.io = best( [ & ] { int i = ::open( src.c_str(), O_RDONLY ), o = ::open( dst.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644 ); double t = IoOnly( i, nullptr, o, nullptr, n, false ); ::close( i ); ::close( o ); return t; } );
            r.ioPre = best( [ & ] { int i = ::open( src.c_str(), O_RDONLY ), o = ::open( dst.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644 ); double t = IoOnly( i, nullptr, o, nullptr, n, true ); ::close( i ); ::close( o ); return t; } );
            r.route = best( [ & ] { const double t0 = Now(); (void)modgpu_cycle_file( src.c_str(), dst.c_str(), kKey, 0, 0 ); return Now() - t0; } );
            rows.push_back( r );
        }
        { // file -> pinned host
            Row r{ "modgpu_cycle_file_to_host (file -> pinned)", 0, 0, 0 };
            r.io = r.ioPre = best( [ & ] { int i = ::open( src.c_str(), O_RDONLY ); double t = IoOnly( i, nullptr, -1, mem, n, false ); ::close( i ); return t; } );
            r.route = best( [ & ] { const double t0 = Now(); (void)modgpu_cycle_file_to_host( src.c_str(), 0, mem, n, kKey, 0, 0 ); return Now() - t0; } );
            rows.push_back( r );
        }
        { // pinned host -> file
            Row r{ "modgpu_cycle_host_to_file (pinned -> file)", 0, 0, 0 };
            r.io = best( [ & ] { int o = ::open( dst.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644 ); double t = IoOnly( -1, mem, o, nullptr, n, false ); ::close( o ); return t; } );
            r.ioPre = best( [ & ] { int o = ::open( dst.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644 ); double t = IoOnly( -1, mem, o, nullptr, n, true ); ::close( o ); return t; } );
            r.route = best( [ & ] { const double t0 = Now(); (void)modgpu_cycle_host_to_file( mem, n, dst.c_str(), kKey, 0, 0 ); return Now() - t0; } );
            rows.push_back( r );
        }
        { // the hypothesis behind the write side's deficit, tested: the same I/O-only job with every chunk evicted from the caches before its pwrite
            double ev = 0;
            const double cold = best( [ & ] { int i = ::open( src.c_str(), O_RDONLY ), o = ::open( dst.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644 ); double t = IoOnly( i, nullptr, o, nullptr, n, true, true, &ev ); ::close( i ); ::close( o ); return t; } );
            std::printf( "   %-44s %8.2f GB/s   (of that call %.0f %% of the wall clock was the eviction itself; without it: %.2f GB/s)\n", "I/O only file -> file, pwrite from DRAM",
                         n / cold / 1e9, 100.0 * ev / cold, n / std::max( cold - ev, 1e-9 ) / 1e9 );
        }
        for( const Row& r : rows )
        {
            const double ceiling = std::max( std::min( r.io, r.ioPre ), gpuOnly ); // the slower of the two stages bounds an overlapped pipeline
            std::printf( "   %-44s %8.2f GB/s   I/O only %7.2f (fallocate first: %7.2f)   -> %3.0f %% of its ceiling (%.2f GB/s)\n", r.name, n / r.route / 1e9,
                         n / r.io / 1e9, n / r.ioPre / 1e9, 100.0 * ceiling / r.route, n / ceiling / 1e9 );
        }
        ::unlink( src.c_str() );
        ::unlink( dst.c_str() );
        TRY( modgpu_host_free( pinned ) );
    }
    return 0;
}

// ---- --alloc: what the part buffer costs to get ----------------------------------------------------------------------------
int Alloc( uint64_t n, int nParts )
{
    std::printf( "== page-locked host memory for a %.0f MiB part buffer in %d parts: allocate, free (seconds; best of 3)\n", n / 1048576.0, nParts );
    auto best = []( auto&& run ) { double b = 1e30; for( int i = 0; i < 3; ++i ) b = std::min( b, run() ); return b; };
    void* p = nullptr;
    double tFree = 0;
    const double a = best( [ & ] { const double t0 = Now(); (void)modgpu_host_alloc( &p, n ); const double t = Now() - t0; const double f0 = Now(); (void)modgpu_host_free( p ); tFree = Now() - f0; return t; } );
    std::printf( "   modgpu_host_alloc       (hipHostMalloc)                                  %.3f s  (%.1f GB/s)   free %.3f s\n", a, n / a / 1e9, tFree );
    std::vector< uint64_t > sizes( (size_t)nParts, n / (uint64_t)nParts );
    sizes.back() += n - sizes[ 0 ] * (uint64_t)nParts;
    const double b = best( [ & ] { const double t0 = Now(); (void)modgpu_host_alloc_parts( &p, sizes.data(), nParts, 0 ); const double t = Now() - t0; const double f0 = Now(); (void)modgpu_host_free( p ); tFree = Now() - f0; return t; } );
    std::printf( "   modgpu_host_alloc_parts (mmap, per-part policy, parallel first touch, lock) %.3f s  (%.1f GB/s)   free %.3f s\n", b, n / b / 1e9, tFree );
    TRY( modgpu_host_alloc_parts( &p, sizes.data(), nParts, 0 ) );
    std::printf( "   placed memory is page-locked: %s; GPU 0 is on NUMA node %d\n", modgpu_host_is_pinned( p, n ) ? "yes" : "NO", modgpu_device_numa_node( 0 ) );
    // is locked-in-place memory as good a target for the kernel across PCIe as hipHostMalloc'ed memory?
    for( int which = 0; which < 2; ++which )
    {
        void* q = nullptr;
        if( which == 0 ) TRY( modgpu_host_alloc( &q, n ) ); else q = p;
        std::memset( q, 0x11, n );
        (void)modgpu_cycle_host( static_cast< uint8_t* >( q ), n, kKey, 0, 0 );
        const double t = best( [ & ] { const double t0 = Now(); (void)modgpu_cycle_host( static_cast< uint8_t* >( q ), n, kKey, 0, 0 ); return Now() - t0; } );
        std::printf( "   modgpu_cycle_host in place on %-24s %.2f GB/s\n", which == 0 ? "modgpu_host_alloc memory" : "placed memory", n / t / 1e9 );
        if( which == 0 ) TRY( modgpu_host_free( q ) );
    }
    TRY( modgpu_host_free( p ) );
    return 0;
}

// ---- --numa: does it matter which socket's memory the kernel works on across PCIe? ---------------------------------------
int Numa( uint64_t n )
{
    const int near = modgpu_device_numa_node( 0 );
    std::printf( "== page-locked memory by NUMA node, %.0f MiB, GPU 0 on node %d: modgpu_cycle_host in place (kernel across PCIe), best of 5; and a 1-thread memset of it\n", n / 1048576.0, near );
    for( int node = 0; node < 8; ++node )
    {
        void* p = nullptr;
        if( modgpu_host_alloc_on_node( &p, n, node ) != MODGPU_OK ) { if( node < 2 ) std::printf( "   node %d: %s\n", node, modgpu_last_error() ); continue; }
        std::memset( p, 0x22, n );
        (void)modgpu_cycle_host( static_cast< uint8_t* >( p ), n, kKey, 0, 0 );
        double best = 1e30, bestSet = 1e30;
        for( int i = 0; i < 5; ++i ) { const double t0 = Now(); (void)modgpu_cycle_host( static_cast< uint8_t* >( p ), n, kKey, 0, 0 ); best = std::min( best, Now() - t0 ); }
        for( int i = 0; i < 3; ++i ) { const double t0 = Now(); std::memset( p, i, n ); bestSet = std::min( bestSet, Now() - t0 ); }
        std::printf( "   node %d%s  kernel across PCIe %6.2f GB/s   pinned: %s   host memset %5.1f GB/s\n", node, node == near ? " (the GPU's)" : "            ", n / best / 1e9,
                     modgpu_host_is_pinned( p, n ) ? "yes" : "no", n / bestSet / 1e9 );
        TRY( modgpu_host_free( p ) );
    }
    return 0;
}

std::vector< int > IntList( const char* s )
{
    std::vector< int > v;
    for( const char* p = s; *p; ) // "0,2,5" or ranges "0..7" (also mixed: "0..3,6")
    {
        const int a = std::atoi( p );
        int b = a;
        while( *p && *p != ',' && *p != '.' ) ++p;
        if( p[ 0 ] == '.' && p[ 1 ] == '.' ) { b = std::atoi( p + 2 ); while( *p && *p != ',' ) ++p; }
        for( int d = a; d <= b && d - a < 64; ++d ) v.push_back( d );
        while( *p && *p != ',' ) ++p;
        if( *p == ',' ) ++p;
    }
    return v;
}
} // namespace

int main( int argc, char** argv )
{
    std::string mode = "classic", dir = "/dev/shm";
    int nParts = 0, steps = 20, warmup = 3;
    uint64_t partBytes = 1ull << 32;
    std::vector< int > devices;
    std::vector< uint64_t > fileSizes;
    std::vector< const char* > positional;
    bool trace = false;
    std::string route;
    uint64_t routeMib = 64;
    int routeReps = 10;
    for( int i = 1; i < argc; ++i )
    {
        const std::string a = argv[ i ];
        auto next = [ & ]() -> const char* { return i + 1 < argc ? argv[ ++i ] : ""; };
        if( a == "--parts" ) { mode = "parts"; nParts = std::atoi( next() ); }
        else if( a == "--devices" ) devices = IntList( next() );
        else if( a == "--part-bytes" ) partBytes = std::strtoull( next(), nullptr, 0 );
        else if( a == "--steps" ) steps = std::atoi( next() );
        else if( a == "--warmup" ) warmup = std::atoi( next() );
        else if( a == "--hostcall" ) mode = "hostcall";
        else if( a == "--trace" ) trace = true;
        else if( a == "--route" ) { mode = "route"; route = next(); }
        else if( a == "--mib" ) routeMib = std::strtoull( next(), nullptr, 0 );
        else if( a == "--offset" ) gRouteOffset = std::strtoull( next(), nullptr, 0 ) & 4095;
        else if( a == "--socket" ) gRouteSocket = next();
        else if( a == "--dir" ) gRouteDir = next();
        else if( a == "--reps" ) routeReps = std::max( 1, std::atoi( next() ) );
        else if( a == "--alloc" ) { mode = "alloc"; partBytes = 3291444381ull; nParts = 8; }
        else if( a == "--numa" ) { mode = "numa"; partBytes = 1ull << 30; }
        else if( a == "--files" ) { mode = "files"; dir = next(); }
        else if( a == "--bytes" ) fileSizes.push_back( std::strtoull( next(), nullptr, 0 ) );
        else positional.push_back( argv[ i ] );
    }
    if( mode == "hostcall" ) return trace ? HostTrace() : HostCall();
    if( mode == "route" ) return route == "pinned" || route == "staged" || route == "file_pageable" || route == "file_pinned" ? Route( route, routeMib, routeReps ) : 1;
    if( modgpu_device_count() < 1 ) { std::printf( "no HIP device\n" ); return 1; }
    if( mode == "parts" ) return nParts > 0 ? Parts( nParts, devices, partBytes, steps, std::max( warmup, 1 ) ) : 1;
    if( mode == "files" ) return Files( dir, fileSizes );
    if( mode == "alloc" ) return Alloc( partBytes, nParts > 0 ? nParts : 8 );
    if( mode == "numa" ) return Numa( partBytes );
    const uint64_t n = positional.size() > 0 ? std::strtoull( positional[ 0 ], nullptr, 0 ) : ( 1ull << 32 );
    return Classic( n, positional.size() > 1 ? std::atoi( positional[ 1 ] ) : 20, positional.size() > 2 ? std::atoi( positional[ 2 ] ) : 3,
                    positional.size() > 3 ? std::atoi( positional[ 3 ] ) : 0 );
}
