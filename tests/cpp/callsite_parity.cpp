// tests/cpp/callsite_parity.cpp -- the reference's three call shapes, compiled against THIS repo's
// CEncryptionCycler.h / Settings.h and checked against the CPU oracle, all in the reference's own
// language.  The bodies of the three blocks follow the call sites they stand for:
//   (1) CArk::Load      Modulate/CArk.cpp:328-339      (2) SaveArk   Modulate/CArk.cpp:914-915, 1133-1136
//   (3) Decode          Modulate/Modulate.cpp:475-486
// Built and run by tests/test_host_gpu.py::test_cpp_callsites (g++, links libmodulate_host.so for the
// product and liboracle_cycle.so for the checker).  Exit code 0 = all equal.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "CEncryptionCycler.h"
#include "Settings.h"
#include "cycle_oracle.h" // TEST INFRASTRUCTURE: the checker

static std::vector< unsigned char > Plain( unsigned int liSize, unsigned int luSeed )
{
    std::vector< unsigned char > v( liSize );
    for( unsigned int ii = 0; ii < liSize; ++ii )
    {
        luSeed = luSeed * 1664525u + 1013904223u;
        v[ ii ] = (unsigned char)( luSeed >> 24 );
    }
    return v;
}

static int Fail( const char* lpWhat, unsigned int liSize )
{
    std::printf( "MISMATCH in %s at size %u\n", lpWhat, liSize );
    return 1;
}

int main()
{
    const unsigned int kaSizes[] = { 4, 5, 19, 20, 21, 4096, 65540, 300007, 524288 };
    for( unsigned int liHeaderSize : kaSizes )
    {
        for( int liPlatform = 0; liPlatform < 2; ++liPlatform )
        {
            const bool lbPS4 = liPlatform == 1;
            std::vector< unsigned char > lImage = Plain( liHeaderSize, liHeaderSize * 7u + liPlatform );

            // (2) save side: magic at 0, then the cipher over the rest with the platform key
            CSettings::SelectPlatform( lbPS4 );
            unsigned char* lacHeaderData = lImage.data();
            const unsigned int kuEncryptedVersion = CSettings::mbPS4 ? CSettings::kuEncryptedVersionPS4 : CSettings::kuEncryptedVersionPS3;
            *(unsigned int*)( lacHeaderData ) = kuEncryptedVersion;
            std::vector< unsigned char > lWant = lImage;
            int liHeaderDataSize = (int)liHeaderSize;
            CEncryptionCycler lEncrypt;
            lEncrypt.Cycle( lacHeaderData + sizeof( unsigned int ), liHeaderDataSize - sizeof( unsigned int ), CSettings::mbPS4 ? CSettings::kuEncryptedPS4Key : CSettings::kuEncryptedPS3Key );
            if( oracle_hdr_encrypt( lWant.data(), liHeaderSize, lbPS4 ? 1 : 0 ) != 0 || lWant != lImage ) return Fail( "SaveArk framing", liHeaderSize );

            // (1) load side: the key follows the file's magic, whatever the platform switch says
            CSettings::SelectPlatform( !lbPS4 );
            unsigned char* lpHeaderData = lImage.data();
            unsigned int luVersion = *(unsigned int*)( lpHeaderData );
            if( luVersion != CSettings::kuEncryptedVersionPS3 && luVersion != CSettings::kuEncryptedVersionPS4 ) return Fail( "magic", liHeaderSize );
            const unsigned int kuInitialKey = ( luVersion == CSettings::kuEncryptedVersionPS3 ) ? CSettings::kuEncryptedPS3Key : CSettings::kuEncryptedPS4Key;
            CEncryptionCycler lDecrypt;
            lDecrypt.Cycle( lpHeaderData + sizeof( unsigned int ), liHeaderSize - sizeof( unsigned int ), kuInitialKey );
            if( oracle_hdr_decrypt( lWant.data(), liHeaderSize ) != 0 || lWant != lImage ) return Fail( "Load framing", liHeaderSize );

            // (3) Decode is the same call on a freshly read file image: one more round trip
            lDecrypt.Cycle( lpHeaderData + sizeof( unsigned int ), liHeaderSize - sizeof( unsigned int ), kuInitialKey );
            oracle_cycle( lWant.data() + 4, liHeaderSize - 4, (int)kuInitialKey );
            if( lWant != lImage ) return Fail( "Decode framing", liHeaderSize );
        }
    }
    // keys the reference passes as `int`: negative, INT_MIN, zero residue (identity)
    const int kaKeys[] = { -1, (int)0x80000000u, 0, 0x7FFFFFFF, (int)0x80000001u, 12345, -127772 };
    for( int liKey : kaKeys )
    {
        std::vector< unsigned char > a = Plain( 70001, (unsigned int)liKey ), b = a;
        CEncryptionCycler c;
        c.Cycle( a.data(), (unsigned int)a.size(), liKey );
        oracle_cycle( b.data(), (uint32_t)b.size(), liKey );
        if( a != b ) return Fail( "key edge cases", (unsigned int)liKey );
    }
    std::printf( "CALLSITES_OK\n" );
    return 0;
}
