// CEncryptionCycler.h -- drop-in for Modulate/CEncryptionCycler.h:3-10.
//
// Same class name, same public signature, stateless and default-constructible, so the
// reference's three callers compile against it unchanged:
//     CArk::Load        Modulate/CArk.cpp:338-339
//     CArk::SaveArk     Modulate/CArk.cpp:1135-1136
//     Decode            Modulate/Modulate.cpp:485-486
// The reference's private CycleKey helper is gone: the step lives in the gfx950 kernel
// (modulate_amd/csrc/cycle_kernel_impl.h) behind the C ABI of include/modgpu.h.
#pragma once

class CEncryptionCycler
{
public:
    // In-place LCG-XOR of liDataSize bytes at lpData (a HOST pointer, any alignment), keystream
    // restarted from liInitialKey -- bit-identical to the reference loop.  Computed on the MI355X;
    // like the reference's, this Cycle cannot fail: on a host with no usable GPU the library's own
    // host loop produces the same bytes (modgpu_cycle_auto_host).  The one exception is opt-in:
    // with MODGPU_REQUIRE_GPU=1 in the environment a missing GPU throws std::runtime_error instead
    // (test-suites and benchmarks set it so that nothing is ever measured on the wrong engine).
    void Cycle( unsigned char* lpData, unsigned int liDataSize, int liInitialKey );

    // Which GPU the calling thread's Cycle calls use (-1 = that thread's current HIP device).
    static void SetDevice( int liDevice );
    static int GetDevice();
};
