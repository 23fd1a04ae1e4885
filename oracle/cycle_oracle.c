/*
 * oracle/cycle_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See cycle_oracle.h.
 *
 * Plain-C restatement of Modulate/CEncryptionCycler.cpp:4-25 written from the
 * description in SURVEY.md 2.1; no reference text is reproduced.
 */
#include "cycle_oracle.h"

#define LCG_M 0x7FFFFFFF /* 2^31 - 1                      */
#define LCG_A 16807      /* 0x41A7                         */
#define LCG_Q 127773     /* 0x1F31D = m / a                */
#define LCG_R 2836       /* 0xB14   = m % a                */

/* CEncryptionCycler.cpp:16-25: Schrage split with C truncating division, one
 * conditional add of m when the result is <= 0. */
int32_t oracle_cycle_key(int32_t key)
{
    int32_t hi = key / LCG_Q;
    int32_t lo = key - hi * LCG_Q;
    int32_t t = lo * LCG_A - hi * LCG_R;
    if (t <= 0)
        t += LCG_M;
    return t;
}

/* CEncryptionCycler.cpp:4-14: prime once, then per byte XOR with (key ^ 0xFF)
 * truncated to 8 bits, step the key. */
void oracle_cycle(uint8_t *buf, uint32_t n, int32_t key0)
{
    int32_t k = oracle_cycle_key(key0);
    for (uint32_t i = 0; i < n; ++i) {
        buf[i] = (uint8_t)(buf[i] ^ (k ^ 0xFF));
        k = oracle_cycle_key(k);
    }
}

void oracle_cycle_serial64(uint8_t *buf, uint64_t n, int32_t key0)
{
    int32_t k = oracle_cycle_key(key0);
    for (uint64_t i = 0; i < n; ++i) {
        buf[i] = (uint8_t)(buf[i] ^ (k ^ 0xFF));
        k = oracle_cycle_key(k);
    }
}

static uint32_t mulmod(uint32_t x, uint32_t y)
{
    return (uint32_t)(((uint64_t)x * y) % LCG_M);
}

static uint32_t powmod(uint32_t base, uint64_t e)
{
    uint32_t r = 1;
    while (e) {
        if (e & 1)
            r = mulmod(r, base);
        base = mulmod(base, base);
        e >>= 1;
    }
    return r;
}

/* SURVEY 2.1 closed form: s_i = a^(i+1) * (key0 mod m) mod m, residue 0 shown as m. */
int32_t oracle_state_at(int32_t key0, uint64_t i)
{
    int64_t k = (int64_t)key0 % LCG_M;
    if (k < 0)
        k += LCG_M;
    uint64_t e = (i % (uint64_t)(LCG_M - 1)) + 1; /* exponent mod the group order */
    uint32_t s = mulmod(powmod(LCG_A, e), (uint32_t)k);
    return s ? (int32_t)s : LCG_M;
}

uint8_t oracle_keystream_at(int32_t key0, uint64_t i)
{
    return (uint8_t)((oracle_state_at(key0, i) & 0xFF) ^ 0xFF);
}

void oracle_cycle_at(uint8_t *buf, uint64_t n, int32_t key0, uint64_t stream_off)
{
    if (!n)
        return;
    int32_t k = oracle_state_at(key0, stream_off);
    for (uint64_t i = 0; i < n; ++i) {
        buf[i] = (uint8_t)(buf[i] ^ (k ^ 0xFF));
        k = oracle_cycle_key(k);
    }
}

uint64_t oracle_fnv1a64(const uint8_t *p, uint64_t n, uint64_t h)
{
    for (uint64_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

static uint32_t rd_le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static void wr_le32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

int oracle_hdr_decrypt(uint8_t *hdr, uint32_t size)
{
    if (size < 4)
        return 6; /* eError_InvalidData: the reference would read out of bounds here */
    uint32_t magic = rd_le32(hdr);
    if (magic != ORACLE_MAGIC_PS3 && magic != ORACLE_MAGIC_PS4)
        return 3; /* eError_UnknownVersionNumber, CArk.cpp:329-334 */
    uint32_t key = (magic == ORACLE_MAGIC_PS3) ? ORACLE_KEY_PS3 : ORACLE_KEY_PS4; /* CArk.cpp:336 */
    oracle_cycle(hdr + 4, size - 4, (int32_t)key);                                /* CArk.cpp:338-339 */
    return 0;
}

int oracle_hdr_encrypt(uint8_t *hdr, uint32_t size, int ps4)
{
    if (size < 4)
        return 6;
    wr_le32(hdr, ps4 ? ORACLE_MAGIC_PS4 : ORACLE_MAGIC_PS3);                     /* CArk.cpp:914-915 */
    oracle_cycle(hdr + 4, size - 4, (int32_t)(ps4 ? ORACLE_KEY_PS4 : ORACLE_KEY_PS3)); /* CArk.cpp:1135-1136 */
    return 0;
}
