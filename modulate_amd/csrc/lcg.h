// lcg.h -- Park-Miller arithmetic shared by the host layer and the gfx950 kernel.
//
// The reference steps the generator one byte at a time in Schrage form
// (Modulate/CEncryptionCycler.cpp:16-25).  Everything here works on the equivalent closed
// form  s_i = a^(i+1) * k mod m   so that any byte's state can be reached
// directly; constexpr so the jump tables are baked into the code object at compile time.
#pragma once
#include <cstdint>

namespace lcg {

constexpr uint32_t M = 0x7FFFFFFFu; // 2^31 - 1 (prime)
constexpr uint32_t A = 16807u;      // 0x41A7, a primitive root of M
constexpr uint32_t PERIOD = M - 1u; // multiplicative order of A

constexpr uint32_t mulmod(uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)x * y) % M); }

constexpr uint32_t powmod(uint32_t b, uint64_t e)
{
    uint32_t r = 1;
    while (e) {
        if (e & 1) r = mulmod(r, b);
        b = mulmod(b, b);
        e >>= 1;
    }
    return r;
}

// key as the reference receives it (int) -> residue in [0, M)
constexpr uint32_t key_residue(int32_t key)
{
    int64_t k = (int64_t)key % (int64_t)M;
    return (uint32_t)(k < 0 ? k + M : k);
}

// State held when stream byte i is XORed, as a residue in [0, M) (0 only if the key is 0 mod M).
constexpr uint32_t state_residue(uint32_t key_res, uint64_t i)
{
    return mulmod(powmod(A, (i % PERIOD) + 1), key_res);
}

// ---- compile-time jump tables --------------------------------------------------------
// Bytes are laid out 16 per lane-word, 256 lane-words (4096 B) per workgroup tile.
constexpr int WORD = 16;        // bytes per lane per access (one dwordx4)
constexpr int BLOCK = 256;      // threads per workgroup (4 waves of 64)
constexpr int TILE = WORD * BLOCK; // 4096 bytes: what one workgroup covers per sub-step

template <int N> struct Table { uint32_t v[N]; };

template <int N> constexpr Table<N> make_pow_table(uint64_t step)
{
    Table<N> t{};
    uint32_t g = powmod(A, step % PERIOD);
    uint32_t x = 1;
    for (int i = 0; i < N; ++i) { t.v[i] = x; x = mulmod(x, g); }
    return t;
}

constexpr Table<16> kBytePow = make_pow_table<16>(1);                    // a^j
constexpr Table<256> kLanePow = make_pow_table<256>(WORD);               // a^(16 t)
constexpr Table<256> kTileLo = make_pow_table<256>(TILE);                // a^(4096 b)
constexpr Table<256> kTileHi = make_pow_table<256>((uint64_t)TILE * 256); // a^(4096*256 b)

} // namespace lcg
