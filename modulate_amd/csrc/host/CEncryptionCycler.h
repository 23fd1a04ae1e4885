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
    // restarted from liInitialKey -- bit-identical to the reference loop.  Which engine computes is the
    // size dispatch SURVEY 8b prescribes (modgpu_cycle_auto_host): buffers below MODGPU_MIN_GPU_BYTES
    // (16 MiB by default -- every header the reference's three callers can pass) on the library's own
    // host loop, larger ones on the MI355X (or wherever MODGPU_HOST_POLICY=fastest prices them faster).
    // Like the reference's, this Cycle cannot fail: on a host with no usable GPU the host loop produces
    // the same bytes at every size.  The one exception is opt-in: with MODGPU_REQUIRE_GPU=1 in the
    // environment every size runs on the kernel and a missing GPU throws std::runtime_error instead
    // (test-suites and benchmarks set it so that nothing is ever measured on the wrong engine).
    void Cycle( unsigned char* lpData, unsigned int liDataSize, int liInitialKey );

    // Which GPU the calling thread's Cycle calls use (-1 = that thread's current HIP device).
    static void SetDevice( int liDevice );
    static int GetDevice();
};
