// host_capi.cpp -- see host_capi.h.
#include "host_capi.h"

#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "CArk.h"
#include "CDtaFile.h"
#include "CEncryptionCycler.h"
#include "Commands.h"
#include "Settings.h"

namespace
{
thread_local std::string t_err;

template < typename F > int Guard( F&& f )
{
    try
    {
        return (int)f();
    }
    catch( const std::exception& e )
    {
        t_err = e.what();
        return -1;
    }
}
CArk* A( void* p ) { return static_cast< CArk* >( p ); }
const CArk* A( const void* p ) { return static_cast< const CArk* >( p ); }
} // namespace

extern "C" {

const char* modhost_last_error( void ) { return t_err.c_str(); }

void modhost_select_platform( int ps4 ) { CSettings::SelectPlatform( ps4 != 0 ); }

void modhost_set_flags( int overwrite_outputs, int ignore_new_files, int pack_all, int verbose )
{
    CSettings::mbOverwriteOutputFiles = overwrite_outputs != 0;
    CSettings::mbIgnoreNewFiles = ignore_new_files != 0;
    CSettings::mbPackAllFiles = pack_all != 0;
    CSettings::mbVerbose = verbose != 0;
}

void modhost_set_fix_quirks( int on ) { CSettings::mbFixReferenceQuirks = on != 0; }

int modhost_cycle_via_class( uint8_t* buf, uint32_t n, int32_t key, int device )
{
    return Guard( [ & ] {
        CEncryptionCycler::SetDevice( device );
        CEncryptionCycler c;
        c.Cycle( buf, n, key );
        return 0;
    } );
}

int modhost_decode( const char* dir ) { return Guard( [ & ] { return Decode( dir ? dir : "" ); } ); }

int modhost_dta_roundtrip( const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* size, char* dump, uint64_t dump_cap )
{
    return Guard( [ & ] {
        CDtaFile f;
        eError e = f.LoadFromMemory( in, (size_t)n );
        if( e != eError_NoError ) return e;
        std::vector< unsigned char > o;
        f.SaveToMemory( o );
        if( size ) *size = o.size();
        if( out && cap ) std::memcpy( out, o.data(), (size_t)std::min< uint64_t >( cap, o.size() ) );
        if( dump && dump_cap )
        {
            std::string d = f.Dump();
            size_t k = (size_t)std::min< uint64_t >( dump_cap - 1, d.size() );
            std::memcpy( dump, d.data(), k );
            dump[ k ] = 0;
        }
        return eError_NoError;
    } );
}

void* modhost_ark_new( void ) { return new CArk(); }
void modhost_ark_free( void* ark ) { delete A( ark ); }
int modhost_ark_load( void* ark, const char* header_path ) { return Guard( [ & ] { return A( ark )->Load( header_path ); } ); }
int modhost_ark_parse_header( void* ark, const uint8_t* image, uint64_t n )
{
    return Guard( [ & ] { return A( ark )->ParseHeader( std::vector< unsigned char >( image, image + n ) ); } );
}
int modhost_ark_load_data( void* ark ) { return Guard( [ & ] { return A( ark )->LoadArkData(); } ); }
int modhost_ark_extract( void* ark, int first, int count, const char* target_dir )
{
    return Guard( [ & ] { return A( ark )->ExtractFiles( first, count, target_dir ); } );
}
int modhost_ark_construct_from_directory( void* ark, const char* input_dir, const void* reference_ark )
{
    return Guard( [ & ] { return A( ark )->ConstructFromDirectory( input_dir, *A( reference_ark ), {} ); } );
}
int modhost_ark_construct_from_table( void* ark, const char* names, const uint32_t* sizes, int n, int n_arks, const char* ark_prefix )
{
    return Guard( [ & ] {
        std::vector< std::string > lNames;
        std::vector< unsigned int > lSizes( sizes, sizes + n );
        const char* p = names;
        for( int i = 0; i < n; ++i )
        {
            lNames.emplace_back( p );
            p += lNames.back().size() + 1;
        }
        return A( ark )->ConstructFromTable( lNames, lSizes, n_arks, ark_prefix );
    } );
}
int modhost_ark_build( void* ark, const char* input_dir ) { return Guard( [ & ] { return A( ark )->BuildArk( input_dir, {} ); } ); }
int modhost_ark_build_from_memory( void* ark, const uint8_t* data, uint64_t n )
{
    return Guard( [ & ] { return A( ark )->BuildArkFromMemory( reinterpret_cast< const char* >( data ), n ); } );
}
int modhost_ark_save( const void* ark, const char* output_dir, const char* header_name )
{
    return Guard( [ & ] { return A( ark )->SaveArk( output_dir, header_name ); } );
}
int modhost_ark_cycle_parts( void* ark, int32_t key, int n_devices ) { return Guard( [ & ] { return A( ark )->CycleArkData( key, n_devices ); } ); }
void modhost_ark_enable_part_cipher( void* ark, int enable, int n_devices ) { A( ark )->EnablePartCipher( enable != 0, n_devices ); }
int modhost_ark_serialise_header( const void* ark, int encrypt, uint8_t* out, uint64_t cap, uint64_t* size )
{
    return Guard( [ & ] {
        std::vector< unsigned char > h;
        eError e = A( ark )->SerialiseHeader( h, encrypt != 0 );
        if( e != eError_NoError ) return e;
        if( size ) *size = h.size();
        if( out && cap ) std::memcpy( out, h.data(), (size_t)std::min< uint64_t >( cap, h.size() ) );
        return eError_NoError;
    } );
}

int modhost_ark_num_files( const void* ark ) { return A( ark )->GetNumFiles(); }
int modhost_ark_num_arks( const void* ark ) { return A( ark )->GetNumArks(); }
uint32_t modhost_ark_ark_size( const void* ark, int i ) { return A( ark )->GetArkSize( i ); }
const char* modhost_ark_ark_path( const void* ark, int i ) { return A( ark )->GetArkPath( i ).c_str(); }
const char* modhost_ark_file_name( const void* ark, int i ) { return A( ark )->GetFileName( i ).c_str(); }
uint32_t modhost_ark_file_size( const void* ark, int i ) { return A( ark )->GetFileSize( i ); }
int64_t modhost_ark_file_offset( const void* ark, int i ) { return A( ark )->GetFileOffset( i ); }
int modhost_ark_file_flags1( const void* ark, int i ) { return A( ark )->GetFileFlags1( i ); }
int modhost_ark_file_flags2( const void* ark, int i ) { return A( ark )->GetFileFlags2( i ); }
uint64_t modhost_ark_data_size( const void* ark ) { return A( ark )->GetArkDataSize(); }
const uint8_t* modhost_ark_data( const void* ark ) { return reinterpret_cast< const uint8_t* >( A( ark )->GetArkData() ); }
int modhost_ark_data_pinned( const void* ark ) { return A( ark )->IsArkDataPinned() ? 1 : 0; }

} // extern "C"
