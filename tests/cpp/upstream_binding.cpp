// tests/cpp/upstream_binding.cpp -- the binding INTEGRATION.md tells an upstream maintainer to add, compiled against
// the UPSTREAM header where it lies (-I/root/reference/Modulate: their pch.h and CEncryptionCycler.h, nothing of
// theirs is copied here) and linked against libmodgpu.so.  tests/test_host_cpu.py builds and runs it when the
// reference tree is present: the class the reference's three call sites use, with this repo's body.
#include "pch.h"
#include "CEncryptionCycler.h"
#include "modgpu.h"
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

void CEncryptionCycler::Cycle( unsigned char* lpData, unsigned int liDataSize, int liInitialKey )
{
    if( modgpu_cycle_auto_host( lpData, liDataSize, liInitialKey, 0, /*device*/ -1 ) != MODGPU_OK )
        throw std::runtime_error( std::string( "Cycle: " ) + modgpu_last_error() );
}

int main()
{
    // the call shape of CArk.cpp:338-339 / 1135-1136 / Modulate.cpp:485-486: Cycle( buf + 4, size - 4, key )
    static const unsigned char kaWant[ 32 ] = { 0x7a, 0xcc, 0xad, 0x6f, 0xaf, 0x91, 0xa7, 0xe3, 0x72, 0x00, 0x8f, 0x07, 0x19, 0xba, 0x34, 0x03,
                                                0xbc, 0x26, 0xc7, 0x12, 0x2a, 0x8d, 0xd1, 0x59, 0x2a, 0xe7, 0xa5, 0xb3, 0xf5, 0x22, 0xb7, 0x3a };
    std::vector< unsigned char > lBuffer( 4 + 100000, 0 );
    CEncryptionCycler lDecrypt;
    lDecrypt.Cycle( lBuffer.data() + 4, (unsigned int)lBuffer.size() - 4, (int)0x90cfc0abu );
    if( std::memcmp( lBuffer.data() + 4, kaWant, 32 ) != 0 || lBuffer[ 0 ] | lBuffer[ 3 ] ) return 1;
    lDecrypt.Cycle( lBuffer.data() + 4, (unsigned int)lBuffer.size() - 4, (int)0x90cfc0abu );
    for( unsigned char c : lBuffer ) if( c ) return 2;
    std::printf( "UPSTREAM_BINDING_OK\n" );
    return 0;
}
