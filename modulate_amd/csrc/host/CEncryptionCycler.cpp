// CEncryptionCycler.cpp -- the whole binding between the reference's cipher seam and the HIP
// layer: one call into include/modgpu.h.  Replaces Modulate/CEncryptionCycler.cpp:4-25.
#include "CEncryptionCycler.h"

#include <stdexcept>
#include <string>

#include "../../../include/modgpu.h"

namespace {
thread_local int t_device = -1;
}

void CEncryptionCycler::SetDevice( int liDevice ) { t_device = liDevice; }
int CEncryptionCycler::GetDevice() { return t_device; }

void CEncryptionCycler::Cycle( unsigned char* lpData, unsigned int liDataSize, int liInitialKey )
{
    // stream offset 0: every Cycle call restarts the keystream (CEncryptionCycler.cpp:7).
    // _auto_: the size dispatch -- header-sized buffers on the library's host loop, larger ones on the
    // GPU kernel, the host loop again where no GPU is usable or the GPU is lost in the middle of the call
    // (the library finishes the pieces that have not arrived): the reference's Cycle returns void and
    // cannot fail, and its callers (CArk.cpp:338-339, 1135-1136, Modulate.cpp:485-486) do not guard it.
    const int liStatus = modgpu_cycle_auto_host( lpData, liDataSize, liInitialKey, 0, t_device );
    if( liStatus != MODGPU_OK )
    {
        // reachable in two ways only, both asked for by the caller: MODGPU_REQUIRE_GPU=1 forbade the host loop, or lpData is
        // page-locked memory (modgpu_host_alloc / _register), which ONE kernel cycles in place -- if that kernel dies under way
        // nobody knows which bytes it wrote.  Failing silently would hand back bytes that are not the reference's.
        throw std::runtime_error( std::string( "CEncryptionCycler::Cycle: GPU path failed: " ) + modgpu_last_error() );
    }
}
