/*
 * oracle/cycle_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's LCG-XOR stream cipher
 * (Modulate/CEncryptionCycler.cpp:4-25, decl Modulate/CEncryptionCycler.h:3-10)
 * plus the header framing the reference wraps around it
 * (Modulate/CArk.cpp:311-339 load, :911-917,1133-1136 save, Modulate/Modulate.cpp:452-502).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * into this library -- as the checker, never as the thing measured or shipped.
 * The product path (modulate_amd/csrc, include/modgpu.h) never links it.
 *
 * Parity status: PINNED for rows a1-a3 (the arithmetic) -- validated against the
 * reference object compiled from /root/reference (oracle/_ref, see oracle/Makefile)
 * and against the golden vectors that build emitted into tests/golden/.
 * Rows a4-a6 (framing) are "parity unpinned": the reference has no tests and its
 * Win32 translation units cannot be built here; the framing below is
 * a restatement of the cited lines only.
 */
#ifndef MODULATE_ORACLE_CYCLE_H
#define MODULATE_ORACLE_CYCLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CEncryptionCycler::CycleKey -- CEncryptionCycler.cpp:16-25. */
int32_t oracle_cycle_key(int32_t key);

/* CEncryptionCycler::Cycle -- CEncryptionCycler.cpp:4-14.  Same 32-bit length. */
void oracle_cycle(uint8_t *buf, uint32_t n, int32_t key0);

/* Same serial loop with a 64-bit length (no jump-ahead: it really steps n times).
 * Used to pin the >= 2^32 cases the reference signature cannot express. */
void oracle_cycle_serial64(uint8_t *buf, uint64_t n, int32_t key0);

/* State the reference loop holds when it XORs byte i (i.e. after i+1 CycleKey calls),
 * obtained by closed form a^(i+1)*key0 mod m.  Value in [1, m]. */
int32_t oracle_state_at(int32_t key0, uint64_t i);

/* Keystream byte i: low8(state_at(i)) ^ 0xFF. */
uint8_t oracle_keystream_at(int32_t key0, uint64_t i);

/* Cycle a window of a longer logical stream: buf[j] ^= ks[stream_off + j].
 * Jumps to stream_off by closed form, then runs the reference's serial step. */
void oracle_cycle_at(uint8_t *buf, uint64_t n, int32_t key0, uint64_t stream_off);

/* FNV-1a 64 used by the golden fixtures. */
uint64_t oracle_fnv1a64(const uint8_t *p, uint64_t n, uint64_t seed);
#define ORACLE_FNV_OFFSET 0xcbf29ce484222325ull

/* Framing restatements.  Return 0 on success, or the reference's eError ordinal
 * (Modulate/Error.h:5-20) -- 3 = eError_UnknownVersionNumber. */
#define ORACLE_MAGIC_PS3 0xc64eed30u /* Settings.h:16 */
#define ORACLE_MAGIC_PS4 0x6f303f55u /* Settings.h:17 */
#define ORACLE_KEY_PS3   0xc64eed30u /* Settings.h:19 */
#define ORACLE_KEY_PS4   0x90cfc0abu /* Settings.h:20 */

/* CArk::Load framing (CArk.cpp:328-339) == Decode (Modulate.cpp:475-486):
 * LE u32 magic at 0 picks the key, Cycle covers [4, size). In place. */
int oracle_hdr_decrypt(uint8_t *hdr, uint32_t size);

/* SaveArk::lSaveHeader framing (CArk.cpp:914-915,1135-1136): writes the magic for the
 * platform at 0, Cycle covers [4, size) with the platform key. In place. */
int oracle_hdr_encrypt(uint8_t *hdr, uint32_t size, int ps4);

#ifdef __cplusplus
}
#endif
#endif
