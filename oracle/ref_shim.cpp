// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE.  C entry point over the *compiled reference*
// class so Python/ctypes can drive it.  Built only by oracle/Makefile target `ref`, which
// compiles /root/reference/Modulate/CEncryptionCycler.cpp where it lies (never copied).
#include <cstdint>
#include "CEncryptionCycler.h" // found via -I/root/reference/Modulate

extern "C" void ref_cycle(unsigned char *buf, unsigned int n, int key)
{
    CEncryptionCycler c;
    c.Cycle(buf, n, key);
}
